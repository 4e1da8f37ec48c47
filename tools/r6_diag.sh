#!/bin/bash
# tools/r6_diag.sh -- round 6 diagnostics in one lease: where the C++ host's time goes on the EM bootstrap job (HIP API
# trace), what ngd_finish costs on this box's cores, and the HIP calls of a cfg 2 bench step.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6
mkdir -p $OUT
cd $ROOT
python3 tools/finish_timing.py > $OUT/finish_timing.txt 2>&1
F=/dev/shm/ngd_e2e_emboot_1000x100000_seed3.bin
[ -f $F ] || tools/gen_gl_file $F 1000 100000 3 16
ARGS="--geno $F --probs --n_ind 1000 --n_sites 100000 --evol_model 2 --n_boot_rep 100 --boot_block_size 10 --out /dev/shm/x.dist --verbose 2 --n_threads 16 --seed 12345"
for i in 1 2 3; do ngsdist_amd/bin/ngsDist $ARGS 2>&1 | grep -E "phases|distances|spill|plan" ; sleep 3; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $OUT/hiptrace_emboot -- $ROOT/ngsdist_amd/bin/ngsDist $ARGS > $OUT/hiptrace_emboot.log 2>&1
grep phases $OUT/hiptrace_emboot.log
head -12 $OUT/hiptrace_emboot/*/*_hip_api_stats.csv
head -8 $OUT/hiptrace_emboot/*/*_kernel_stats.csv | cut -c1-200
rocprofv3 --hip-trace --stats --output-format csv -d $OUT/hiptrace_cfg2 -- python3 $ROOT/bench.py --workload cfg2 --steps 200 --warmup 5 --no_cpu > $OUT/hiptrace_cfg2.json 2> $OUT/hiptrace_cfg2.err
head -14 $OUT/hiptrace_cfg2/*/*_hip_api_stats.csv
cd $ROOT; python3 bench.py --workload cfg2 --steps 200 --warmup 5 --no_cpu > $OUT/cfg2_plain.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('$OUT/cfg2_plain.json').read().strip().splitlines()[-1]); print('cfg2 ms_per_step', d['ms_per_step'], d['roofline']['ms_per_launch'], d['valid'])"
cat $OUT/finish_timing.txt | tail -6
