#!/bin/bash
# tools/em_pmc.sh [kernel] [n_sites] [workload] -- VALU issue / lane occupancy / LDS counters of an EM kernel (cfg4 shape;
# workload emboot: the SPILL instantiation inside the spilled-terms bootstrap job, one launch per chunk of sites)
#   -> gpurun_out/prof_em_<kernel>[_<workload>]/ ; two --pmc passes (8 SQ slots each) + a --kernel-trace --stats pass
set -u
KERNEL=${1:-em_table}
NS=${2:-100000}
WL=${3:-cfg4}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_em_$KERNEL
[ "$WL" = cfg4 ] || OUT=${OUT}_$WL
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $WL --kernel $KERNEL --n_sites $NS --steps 2 --warmup 1 --no_cpu"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/pmc" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc.json" 2> "$OUT/pmc.err"
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAVES \
  --output-format csv -d "$OUT/pmc2" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc2.json" 2> "$OUT/pmc2.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/kt.json" 2> "$OUT/kt.err"
cd "$ROOT" && python3 - "$OUT" "$KERNEL" "$NS" "$WL" <<'PY'
import csv, glob, sys, collections, json
out, kernel, ns, wl = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
d = collections.defaultdict(list)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_accum_em" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in d.items()}
line = json.loads(open(out + "/kt.json").read().strip().splitlines()[-1])
ps = 499500 * ns / line["roofline"].get("launches_per_job", 1)  # pair-sites of ONE launch (emboot: a chunk of sites)
ms = line["roofline"]["ms_per_launch"]
cyc = m["GRBM_GUI_ACTIVE"] / 8
print("# rocprofv3 --pmc, EM kernel %s, workload %s, 1000 x %g sites, one launch = %.4g pair-sites\n" % (kernel, wl, ns, ps))
print("| counter | mean per launch |\n|---|---|")
for k in sorted(m):
    print("| %s | %.4g |" % (k, m[k]))
print("\n- kernel time %.1f ms (HIP events), shader clock %.0f MHz (GRBM_GUI_ACTIVE / 8 / time)" % (ms, cyc / ms / 1e3))
print("- VALU issue: SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x %.3g cycles) = %.2f of all issue slots (at 4 cycles per instruction)" % (cyc, m["SQ_INSTS_VALU"] * 4 / (1024 * cyc)))
print("- lane occupancy: SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = %.1f of 64 lanes (%.2f)" % (
    m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"], m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"] / 64))
print("- VALU wavefront-instructions per pair-site: %.3f (x64 lanes = %.1f lane-slots); active lane-cycles per pair-site (SQ_THREAD_CYCLES_VALU): %.1f" % (
    m["SQ_INSTS_VALU"] / ps, m["SQ_INSTS_VALU"] * 64 / ps, m["SQ_THREAD_CYCLES_VALU"] / ps))
if "SQ_INSTS_LDS" in m:
    print("- LDS: %.3f wavefront-instructions per pair-site; SQ_LDS_IDX_ACTIVE / (256 CUs x cycles) = %.2f; bank-conflict cycles / LDS active = %.3f" % (
        m["SQ_INSTS_LDS"] / ps, m.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc), m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 1))))
print("- wave-cycles: WAIT_ANY %.1f%%, WAIT_INST_ANY %.1f%% of SQ_WAVE_CYCLES; waves per launch %.4g" % (
    100 * m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"], 100 * m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m.get("SQ_WAVES", 0)))
print("- %.3g pair-sites/s" % (ps / (ms * 1e-3)))
# what bench.py's EM roofline reads (profiles/valu_cfg4_<kernel>.json): work per pair-site, tied to the kernel source
import hashlib, os
src = "accum_%s.hip" % {"em_fast": "em", "em_faithful": "em", "em_table": "em_table"}.get(kernel, kernel)
sha = hashlib.sha256(open(os.path.join("ngsdist_amd", "csrc", src), "rb").read()).hexdigest()[:16]
json.dump({"source": "tools/em_pmc.sh %s %g %s (rocprofv3 --pmc, two passes)" % (kernel, ns, wl), "kernel_source_sha16": {src: sha},
           "per_pair_site": {"active_lane_instructions": m["SQ_THREAD_CYCLES_VALU"] / ps, "issue_slots": m["SQ_INSTS_VALU"] * 64 / ps,
                             "lds_wave_instructions": m.get("SQ_INSTS_LDS", 0) / ps},
           "lane_occupancy": m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"] / 64,
           "lds_busy": m.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc), "shader_mhz": cyc / ms / 1e3},
          open(out + "/valu_%s_%s.json" % (wl, kernel), "w"), indent=1)
PY
