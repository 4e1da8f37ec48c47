#!/bin/bash
# tools/final_lines.sh -- the bench lines committed under profiles/<round>_bench_lines/ (run on the GPU box)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/bench_lines
mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py --workload cfg3                                  > "$OUT/cfg3_mfma.json"   2> "$OUT/cfg3_mfma.err"   && echo cfg3 done &&
python3 bench.py --workload cfg2 --steps 50 --warmup 5            > "$OUT/cfg2.json"        2> "$OUT/cfg2.err"        && echo cfg2 done &&
python3 bench.py --workload cfg5 --steps 10 --warmup 3            > "$OUT/cfg5_boot.json"   2> "$OUT/cfg5_boot.err"   && echo cfg5 done &&
python3 bench.py --workload cfg4 --steps 3 --warmup 1             > "$OUT/cfg4_em_table.json" 2> "$OUT/cfg4_em_table.err" && echo cfg4 done &&
python3 bench.py --workload cfg4 --kernel em_fast --steps 2 --warmup 1 --no_cpu > "$OUT/cfg4_em_fast.json" 2> "$OUT/cfg4_em_fast.err" && echo cfg4 em_fast done &&
python3 bench.py --workload cfg3 --kernel stream --steps 2 --warmup 1 --no_cpu > "$OUT/cfg3_stream.json" 2> "$OUT/cfg3_stream.err" && echo stream done
