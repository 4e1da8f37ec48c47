#!/bin/bash
# tools/r4_measure.sh -- round 4's evidence in one lease of the GPU box: bench lines, rocprofv3 passes of the headline
# command (with the unprofiled line it must reconcile with), counters of the EM kernel, the EM bootstrap job.
# Everything lands under gpurun_out/; what is judged is copied to profiles/ by hand.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r4_lines
mkdir -p "$OUT"
cd "$ROOT"
line() { # name, bench args...
  local name=$1; shift
  python3 bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name: exit $?"
}
line cfg3 --workload cfg3
line cfg2 --workload cfg2 --steps 50 --warmup 5
line cfg5 --workload cfg5 --steps 10 --warmup 3
line cfg4 --workload cfg4 --steps 3 --warmup 1
line cfg3_stream --workload cfg3 --kernel stream --steps 1 --warmup 1 --no_cpu --serial_tail
line cfg3_driver_style --workload cfg3 --gpus 1 --steps 20 --warmup 5
tools/profile.sh r4_cfg3 --workload cfg3 > "$OUT/profile_cfg3.log" 2>&1; echo "profile cfg3 done"
tools/em_pmc.sh em_table 100000 > "$OUT/em_pmc.md" 2> "$OUT/em_pmc.err"; echo "em_pmc done"
PASSES="kt fetch write l2" tools/profile.sh r4_cfg4 --workload cfg4 --n_sites 100000 --steps 2 --warmup 1 > "$OUT/profile_cfg4.log" 2>&1; echo "profile cfg4 done"
PASSES="kt fetch write l2 sq" tools/profile.sh r4_cfg2 --workload cfg2 --steps 50 --warmup 5 > "$OUT/profile_cfg2.log" 2>&1; echo "profile cfg2 done"
python3 tools/em_boot_job.py 100000 10 2 5 15 100 > "$OUT/em_boot_job.txt" 2>&1; echo "em_boot_job done"
python3 tools/em_boot_job.py 100000 1 5 100 --no_batch > "$OUT/em_boot_job_block1.txt" 2>&1; echo "em_boot_job block 1 done"
