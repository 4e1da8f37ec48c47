#!/bin/bash
# tools/r6_hiptrace_cfg3.sh -- the C++ host on cfg 3 four times back to back under rocprofv3 --hip-trace (HIP API durations):
# which call holds the time of the runs whose engine creation is slow?  Needs the file of tools/bench_e2e.py --keep.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
F=/dev/shm/ngd_e2e_cfg3_1000x1000000_seed3.bin
[ -f $F ] || $ROOT/tools/gen_gl_file $F 1000 1000000 3 16
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4; do
  rocprofv3 --hip-trace --stats --output-format csv -d $ROOT/gpurun_out/r6/hiptrace_cfg3_$i -- $ROOT/ngsdist_amd/bin/ngsDist --geno $F --probs \
    --n_ind 1000 --n_sites 1000000 --evol_model 1 --indep_geno --out /tmp/x.dist --verbose 2 --n_threads 16 > $ROOT/gpurun_out/r6/hiptrace_cfg3_$i.log 2>&1
  grep phases $ROOT/gpurun_out/r6/hiptrace_cfg3_$i.log
done
