#!/bin/bash
# tools/r4_measure.sh -- round 4's evidence in one lease of the GPU box: bench lines, rocprofv3 passes of the headline
# command (with the unprofiled line it must reconcile with), counters of the EM kernel, the EM bootstrap job.
# Everything lands under gpurun_out/; what is judged is copied to profiles/ by hand.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r4_lines
mkdir -p "$OUT"
cd "$ROOT"
line() { # name, bench args...
  local name=$1; shift
  python3 bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name: exit $?"
}
# (the summary is made HERE, on the box, from this run's files alone: gpurun merges into whatever older runs left locally)
summarize() { python3 tools/pmc_summary.py "$ROOT/gpurun_out/prof_$1" > "$ROOT/gpurun_out/prof_$1/summary.md" 2> "$ROOT/gpurun_out/prof_$1/summary.err"; }
PART=${1:-all}
# (lines: the bench lines alone, once the counter passes they cite -- profiles/traffic_*.json, valu_*.json -- are in place)
if [ "$PART" = lines ]; then
line cfg3 --workload cfg3
line cfg2 --workload cfg2 --steps 50 --warmup 5
line cfg5 --workload cfg5 --steps 10 --warmup 3
line cfg4 --workload cfg4 --steps 3 --warmup 1
line cfg3_stream --workload cfg3 --kernel stream --steps 1 --warmup 1 --no_cpu --serial_tail
line cfg3_driver_style --workload cfg3 --gpus 1 --steps 20 --warmup 5
line cfg3_single_image2 --workload cfg3 --single_image 2 --no_cpu
line cfg3_single_image1 --workload cfg3 --single_image 1 --no_cpu
fi
if [ "$PART" = all ] || [ "$PART" = indep ]; then
line cfg3 --workload cfg3
line cfg2 --workload cfg2 --steps 50 --warmup 5
line cfg5 --workload cfg5 --steps 10 --warmup 3
line cfg3_stream --workload cfg3 --kernel stream --steps 1 --warmup 1 --no_cpu --serial_tail
line cfg3_driver_style --workload cfg3 --gpus 1 --steps 20 --warmup 5
line cfg3_single_image2 --workload cfg3 --single_image 2 --no_cpu
line cfg3_single_image1 --workload cfg3 --single_image 1 --no_cpu
tools/profile.sh r4_cfg3 --workload cfg3 > "$OUT/profile_cfg3.log" 2>&1; summarize r4_cfg3; echo "profile cfg3 done"
PASSES="kt fetch write l2 sq" tools/profile.sh r4_cfg2 --workload cfg2 --steps 50 --warmup 5 > "$OUT/profile_cfg2.log" 2>&1; summarize r4_cfg2; echo "profile cfg2 done"
PASSES="kt fetch write l2 sq" tools/profile.sh r4_cfg5 --workload cfg5 --steps 10 --warmup 3 > "$OUT/profile_cfg5.log" 2>&1; summarize r4_cfg5; echo "profile cfg5 done"
fi
if [ "$PART" = all ] || [ "$PART" = em ]; then
line cfg4 --workload cfg4 --steps 3 --warmup 1
tools/em_pmc.sh em_table 100000 > "$OUT/em_pmc.md" 2> "$OUT/em_pmc.err"; echo "em_pmc done"
PASSES="kt fetch write l2" tools/profile.sh r4_cfg4 --workload cfg4 --n_sites 100000 --steps 2 --warmup 1 > "$OUT/profile_cfg4.log" 2>&1; summarize r4_cfg4; echo "profile cfg4 done"
python3 tools/em_boot_job.py 100000 10 2 5 15 100 > "$OUT/em_boot_job.txt" 2>&1; echo "em_boot_job done"
python3 tools/em_boot_job.py 100000 1 5 100 --no_batch > "$OUT/em_boot_job_block1.txt" 2>&1; echo "em_boot_job block 1 done"
fi
