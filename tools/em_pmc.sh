#!/bin/bash
# tools/em_pmc.sh -- VALU issue / lane occupancy counters of the EM kernel (cfg4 shape, 1e5 sites) -> gpurun_out/prof_em/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_em
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/pmc" -- python3 "$ROOT/bench.py" --workload cfg4 --n_sites 100000 --steps 2 --warmup 1 --no_cpu > "$OUT/pmc.json" 2> "$OUT/pmc.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$ROOT/bench.py" --workload cfg4 --n_sites 100000 --steps 2 --warmup 1 --no_cpu > "$OUT/kt.json" 2> "$OUT/kt.err"
cd "$ROOT" && python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
d = collections.defaultdict(list)
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_accum_em" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in d.items()}
line = json.loads(open(out + "/kt.json").read().strip().splitlines()[-1])
ps = 499500 * 100000.0
ms = line["roofline"]["ms_per_launch"]
cyc = m["GRBM_GUI_ACTIVE"] / 8
print("# rocprofv3 --pmc, EM kernel (k_accum_em<fast>), 1000 x 1e5 sites, one launch = 4.995e10 pair-sites\n")
print("| counter | mean per launch |\n|---|---|")
for k in sorted(m):
    print("| %s | %.4g |" % (k, m[k]))
print("\n- kernel time %.1f ms (HIP events), shader clock %.0f MHz (GRBM_GUI_ACTIVE / 8 / time)" % (ms, cyc / ms / 1e3))
print("- VALU issue: SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x %.3g cycles) = %.2f of all issue slots" % (cyc, m["SQ_INSTS_VALU"] * 4 / (1024 * cyc)))
print("- lane occupancy: SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = %.1f of 64 lanes (%.2f)" % (
    m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"], m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"] / 64))
print("- useful lane-instructions per pair-site: %.0f; issued wavefront-instruction lanes per pair-site: %.0f" % (
    m["SQ_THREAD_CYCLES_VALU"] / ps, m["SQ_INSTS_VALU"] * 64 / ps))
print("- %.3g pair-sites/s" % (ps / (ms * 1e-3)))
PY
