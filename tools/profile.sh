#!/bin/bash
# tools/profile.sh <tag> [bench args...] -- rocprofv3 evidence for one bench.py command:
#   pass 0: an UNPROFILED bench line, then --kernel-trace --stats (per-kernel durations).  The trace is kept only if
#           its own bench line's ms_per_step is within 2 % of the unprofiled line taken in the same lease; otherwise
#           both are taken once more (boxes differ, and a box's first runs can sit at a lower clock) and the better
#           pair is kept -- reconcile.txt says which, and tools/pmc_summary.py prints it
#   pass 1..: one --pmc group per run       (HBM bytes, SQ/MFMA activity, L2/L1 hit rates)
# Output under gpurun_out/prof_<tag>/; tools/pmc_summary.py condenses it for profiles/.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, rocprof args...
  local name=$1; shift
  rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" "${BENCH_ARGS[@]}" \
      > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "pass $name failed (see $OUT/$name.err)"
  echo "pass $name done"
}
BENCH_ARGS=("$@" --no_cpu)
PASSES=${PASSES:-"kt fetch write sq l2 l1"}
for p in $PASSES; do
  case $p in
    kt)
      python3 "$ROOT/bench.py" "${BENCH_ARGS[@]}" > "$OUT/plain.json" 2> "$OUT/plain.err" || echo "unprofiled line failed"
      run kt --kernel-trace --stats
      if ! python3 "$ROOT/tools/reconcile.py" "$OUT/plain.json" "$OUT/kt.json" > "$OUT/reconcile.txt"; then
        mv "$OUT/kt" "$OUT/kt_first"; mv "$OUT/kt.json" "$OUT/kt_first.json"; mv "$OUT/plain.json" "$OUT/plain_first.json"
        python3 "$ROOT/bench.py" "${BENCH_ARGS[@]}" > "$OUT/plain.json" 2> "$OUT/plain.err" || echo "unprofiled line failed"
        run kt --kernel-trace --stats
        { echo "first attempt:"; cat "$OUT/reconcile.txt"; echo "second attempt (kept):";
          python3 "$ROOT/tools/reconcile.py" "$OUT/plain.json" "$OUT/kt.json"; } > "$OUT/reconcile2.txt"
        mv "$OUT/reconcile2.txt" "$OUT/reconcile.txt"
      fi
      cat "$OUT/reconcile.txt" ;;
    fetch) run fetch --kernel-trace --pmc FETCH_SIZE ;;
    write) run write --kernel-trace --pmc WRITE_SIZE ;;
    sq) run sq --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE ;;
    l2) run l2 --kernel-trace --pmc TCC_HIT TCC_MISS ;;
    l1) run l1 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ ;;
  esac
done
cd "$ROOT" && python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.md" && cat "$OUT/summary.md"
