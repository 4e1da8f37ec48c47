#!/bin/bash
# tools/r6_measure.sh [e2e|counters|lines] -- round 6's evidence, one part per lease of the GPU box.
#   e2e:      the C++ host end to end on every BASELINE configuration from a file in host memory (tools/bench_e2e.py: the
#             link's roof by tools/pcie_peak in the same lease, phases, a full check of the printed matrices), the load
#             phase under rocprofv3 (tools/r6_load_trace.sh: kernel + memory-copy trace, FETCH_SIZE / WRITE_SIZE of K0),
#             and the host-side pipelines side by side (tools/host_read_pipeline)
#   counters: tools/profile.sh per workload (kernel trace reconciled against an unprofiled line of the same lease)
#   lines:    the bench lines committed under profiles/r06_bench_lines/
#   tail:     bootstrap jobs through ngd_run_mult_batch_dist (the job and its tail in one call) against the round-5 form of the
#             tail (--split_tail), interleaved in one lease, then the bootstrap workloads' lines again
# Everything lands under gpurun_out/; what is judged is copied to profiles/ by hand.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r6_lines
mkdir -p "$OUT" "$ROOT/gpurun_out/r6"
cd "$ROOT"
line() { # name, bench args...
  local name=$1; shift
  python3 bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name: exit $?"
}
PART=${1:-lines}
if [ "$PART" = e2e ]; then
  python3 tools/bench_e2e.py --workloads cfg3,cfg4,cfg5,emboot,cfg2 --runs 5 --gap 3 --keep > "$OUT/e2e.jsonl" 2> "$OUT/e2e.err"; echo "e2e: exit $?"
  python3 tools/bench_e2e.py --workloads cfg3 --runs 5 --gap 0 --keep --no_roof --no_check > "$OUT/e2e_cfg3_back_to_back.jsonl" 2>> "$OUT/e2e.err"; echo "e2e back to back: exit $?"
  tools/r6_load_trace.sh r06 > "$OUT/load_trace.log" 2>&1; echo "load trace done"
  tools/host_read_pipeline /dev/shm/ngd_e2e_cfg3_1000x1000000_seed3.bin 8 32 6 > "$OUT/host_read_pipeline.json" 2> "$OUT/host_read_pipeline.err"; echo "pipelines done"
  tools/alloc_cost 32 > "$OUT/alloc_cost.json" 2> "$OUT/alloc_cost.err"; echo "alloc cost done"
  rm -f /dev/shm/ngd_e2e_*
fi
if [ "$PART" = counters ]; then
  tools/profile.sh r06_cfg3 --workload cfg3 > "$OUT/profile_cfg3.log" 2>&1; echo "profile cfg3 done"
  PASSES="kt fetch write l2 sq" tools/profile.sh r06_cfg2 --workload cfg2 --steps 50 --warmup 5 > "$OUT/profile_cfg2.log" 2>&1; echo "profile cfg2 done"
  PASSES="kt fetch write l2 sq" tools/profile.sh r06_cfg5 --workload cfg5 --steps 10 --warmup 3 > "$OUT/profile_cfg5.log" 2>&1; echo "profile cfg5 done"
  PASSES="kt fetch write" tools/profile.sh r06_emboot --workload emboot --steps 5 --warmup 2 > "$OUT/profile_emboot.log" 2>&1; echo "profile emboot done"
  PASSES="kt fetch write l2" tools/profile.sh r06_cfg4 --workload cfg4 --n_sites 100000 --steps 2 --warmup 1 > "$OUT/profile_cfg4.log" 2>&1; echo "profile cfg4 done"
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
  line cfg3_two_images --workload cfg3 --single_image 3 --no_cpu
  python3 tools/ab_lines.py "$OUT"/cfg*.json "$OUT"/emboot*.json
fi
if [ "$PART" = tail ]; then
  for i in 1 2 3; do
    line tail_cfg5_one_$i --workload cfg5 --no_cpu --serial_tail
    line tail_cfg5_split_$i --workload cfg5 --no_cpu --serial_tail --split_tail
  done
  for i in 1 2; do
    line tail_emboot_one_$i --workload emboot --no_cpu --serial_tail
    line tail_emboot_split_$i --workload emboot --no_cpu --serial_tail --split_tail
  done
  python3 - "$OUT" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/tail_*.json")):
    j = json.loads(open(f).read())
    print("%s: step %.3f ms, dominant kernel %.3f ms, valid %s" % (f.split("/")[-1], j["ms_per_step"], j["roofline"]["ms_per_launch"], j["valid"]))
PY
  line cfg5 --workload cfg5 --steps 10 --warmup 3
  line emboot --workload emboot
  line emboot_5_replicates --workload emboot --n_boot 5 --no_cpu
  line emboot_block1 --workload emboot --block 1 --no_cpu
  python3 tools/ab_lines.py "$OUT"/cfg5.json "$OUT"/emboot*.json
fi
