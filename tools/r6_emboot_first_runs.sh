#!/bin/bash
# tools/r6_emboot_first_runs.sh -- why the EM bootstrap command's first run or two after other work on the device take 2.8 s
# instead of 0.8: the device's memory is made busy (tools/alloc_cost touches 64 GiB), then the command runs four times, 3 s
# apart, with NGD_TRACE_ALLOC=1 (every device allocation of 64 MiB and more with its duration on stderr).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6/emboot_first
mkdir -p "$OUT"; cd "$ROOT"
F=/dev/shm/ngd_first_emboot.bin
tools/gen_gl_file $F 1000 100000 3 16
tools/alloc_cost 64 > "$OUT/alloc_cost.json" 2> "$OUT/alloc_cost.err"
for i in 1 2 3 4; do
  rm -f /dev/shm/ngd_first_emboot.dist
  s=$(date +%s.%N)
  NGD_TRACE_ALLOC=1 ngsdist_amd/bin/ngsDist --geno $F --probs --n_ind 1000 --n_sites 100000 --evol_model 2 --out /dev/shm/ngd_first_emboot.dist \
      --verbose 2 --n_threads 16 --seed 12345 --n_boot_rep 100 --boot_block_size 10 2> "$OUT/run$i.err" > /dev/null
  e=$(date +%s.%N)
  python3 -c "print('run $i: %.3f s' % ($e - $s))"; grep "alloc:\|phases" "$OUT/run$i.err" | cut -c1-400
  sleep 3
done
rm -f $F /dev/shm/ngd_first_emboot.dist
