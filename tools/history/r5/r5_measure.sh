#!/bin/bash
# tools/r5_measure.sh [counters|lines] -- round 5's evidence in one lease of the GPU box.
#   counters: the rocprofv3 passes bench.py's roofline lines cite (profiles/traffic_*.json, valu_*.json are tied to the
#             kernel sources by sha, so they are taken on the final sources) -- tools/profile.sh per workload (an
#             unprofiled line first, the kernel trace kept only if it reconciles), tools/em_pmc.sh for the EM kernel;
#   lines:    the bench lines committed under profiles/r05_bench_lines/, once those files are in place.
# Everything lands under gpurun_out/; what is judged is copied to profiles/ by hand.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r5_lines
mkdir -p "$OUT"
cd "$ROOT"
line() { # name, bench args...
  local name=$1; shift
  python3 bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name: exit $?"
}
PART=${1:-lines}
if [ "$PART" = counters ]; then
  tools/profile.sh r05_cfg3 --workload cfg3 > "$OUT/profile_cfg3.log" 2>&1; echo "profile cfg3 done"
  PASSES="kt fetch write l2 sq" tools/profile.sh r05_cfg2 --workload cfg2 --steps 50 --warmup 5 > "$OUT/profile_cfg2.log" 2>&1; echo "profile cfg2 done"
  PASSES="kt fetch write l2 sq" tools/profile.sh r05_cfg5 --workload cfg5 --steps 10 --warmup 3 > "$OUT/profile_cfg5.log" 2>&1; echo "profile cfg5 done"
  tools/profile.sh r05_emboot --workload emboot --steps 5 --warmup 2 > "$OUT/profile_emboot.log" 2>&1; echo "profile emboot done"
  tools/em_pmc.sh em_table 100000 > "$OUT/em_pmc_cfg4.md" 2> "$OUT/em_pmc_cfg4.err"; echo "em_pmc cfg4 done"
  tools/em_pmc.sh em_table 100000 emboot > "$OUT/em_pmc_emboot.md" 2> "$OUT/em_pmc_emboot.err"; echo "em_pmc emboot done"
  PASSES="kt fetch write l2" tools/profile.sh r05_cfg4 --workload cfg4 --n_sites 100000 --steps 2 --warmup 1 > "$OUT/profile_cfg4.log" 2>&1; echo "profile cfg4 done"
  PASSES="kt fetch write l2" tools/profile.sh r05_cfg3_two_images --workload cfg3 --single_image 3 > "$OUT/profile_cfg3_two_images.log" 2>&1; echo "profile cfg3 (two images) done"
fi
if [ "$PART" = lines ]; then
  line cfg3 --workload cfg3
  line cfg3_driver_style --workload cfg3 --gpus 1 --steps 20 --warmup 5
  line cfg2 --workload cfg2 --steps 50 --warmup 5
  line cfg5 --workload cfg5 --steps 10 --warmup 3
  line cfg4 --workload cfg4 --steps 3 --warmup 1
  line emboot --workload emboot
  line emboot_5_replicates --workload emboot --n_boot 5 --no_cpu
  line emboot_block1 --workload emboot --block 1 --no_cpu
  line emboot_block1_5_replicates --workload emboot --block 1 --n_boot 5 --no_cpu
  line cfg3_stream --workload cfg3 --kernel stream --steps 1 --warmup 1 --no_cpu --serial_tail
  line cfg3_two_images --workload cfg3 --single_image 3 --no_cpu
  line cfg3_single_image1 --workload cfg3 --single_image 1 --no_cpu
  python3 tools/ab_lines.py "$OUT"/cfg*.json "$OUT"/emboot*.json
fi
