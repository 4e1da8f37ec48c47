#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into a small markdown summary
(per-kernel average duration and per-launch counter values)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]


print("# rocprofv3 summary: %s\n" % os.path.basename(out))
for j in sorted(glob.glob(os.path.join(out, "kt.json"))):
    try:
        d = json.loads(open(j).read().strip().splitlines()[-1])
        print("bench line of the kernel-trace pass: value=%.4g %s, ms_per_step=%.3f, roofline=%s\n"
              % (d["value"], d["unit"], d["ms_per_step"], json.dumps({k: d["roofline"][k] for k in
                                                                         ("kernel", "achieved", "peak", "unit", "frac", "ms_per_launch")})))
        print("workload: %s\n" % d["config"]["workload"])
    except Exception as exc:
        print("(no bench line: %r)\n" % exc)

if os.path.exists(os.path.join(out, "reconcile.txt")):
    print("## the traced run against an unprofiled line of the same lease (tools/reconcile.py)\n")
    print("```\n%s```\n" % open(os.path.join(out, "reconcile.txt")).read())

line = None
try:
    line = json.loads(open(os.path.join(out, "kt.json")).read().strip().splitlines()[-1])
except Exception:
    pass
stats = glob.glob(os.path.join(out, "kt", "**", "*kernel_stats.csv"), recursive=True)
kt_ms = {}
if stats:
    for r in csv.DictReader(open(stats[0])):
        kt_ms[short(r["Name"])] = float(r["AverageNs"]) / 1e6
    print("## --kernel-trace --stats\n")
    print("| kernel | calls | avg ms | min ms | max ms | % |")
    print("|---|---|---|---|---|---|")
    for r in csv.DictReader(open(stats[0])):
        print("| %s | %s | %.4f | %.4f | %.4f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e6,
                                                         float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Percentage"]))
    print()
    if line and line["roofline"].get("frac") is not None:
        r = line["roofline"]
        for k, ms in kt_ms.items():
            if r["kernel"] in k and "reduce" not in k:
                print("`%s`: trace average %.4f ms against %.4f ms by HIP events inside the same run (ms_per_step %.4f) -> "
                      "roofline.frac recomputed from the trace = %.4f (the line says %.4f)\n"
                      % (k, ms, r["ms_per_launch"], line["ms_per_step"], r["frac"] * r["ms_per_launch"] / ms, r["frac"]))
                break
        c = r.get("contract")
        if c:
            for k, ms in kt_ms.items():
                if "k_contract_mfma" in k:
                    print("`%s`: trace average %.4f ms against %.4f ms per launch by HIP events inside the same run -> its "
                          "roofline.contract.frac recomputed from the trace = %.4f (the line says %.4f, bound: %s)\n"
                          % (k, ms, c["ms_per_launch"], c["frac"] * c["ms_per_launch"] / ms, c["frac"], c["bound"]))

print("## --pmc passes (mean per launch; every pass is a separate run)\n")
print("| pass | kernel | counter | launches | mean per launch |")
print("|---|---|---|---|---|")
agg = {}
for p in ("fetch", "write", "sq", "l2", "l1"):
    files = glob.glob(os.path.join(out, p, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        acc = defaultdict(lambda: [0.0, set()])
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), r["Counter_Name"])
            acc[key][0] += float(r["Counter_Value"])
            acc[key][1].add(r["Dispatch_Id"])
        for (k, c), (v, ids) in sorted(acc.items()):
            if not any(w in k for w in ("accum", "reduce", "count", "contract", "spill")):
                continue
            mean = v / max(1, len(ids))
            agg[(k, c)] = mean
            print("| %s | %s | %s | %d | %.6g |" % (p, k, c, len(ids), mean))
print()
traffic = {}
print("## derived (accumulation kernels; k_contract_mfma of the spilled-terms plan)\n")
for k in sorted({k for (k, _) in agg}):
    if "accum" not in k and "contract" not in k:
        continue
    g = lambda c: agg.get((k, c))
    print("kernel `%s`:" % k)
    if g("FETCH_SIZE") is not None:
        # rocprofv3 reports KiB; MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reads exactly half of the
        # bytes of a wide coalesced streaming read -> double it; WRITE_SIZE is exact.
        fb = g("FETCH_SIZE") * 1024 * 2
        print("- HBM read bytes per launch  = FETCH_SIZE x 1024 x 2 (gfx950 correction) = %.4g GB" % (fb / 1e9))
        traffic.setdefault(k, {})["read_bytes"] = fb
    if g("WRITE_SIZE") is not None:
        print("- HBM write bytes per launch = WRITE_SIZE x 1024 = %.4g GB" % (g("WRITE_SIZE") * 1024 / 1e9))
        traffic.setdefault(k, {})["write_bytes"] = g("WRITE_SIZE") * 1024
    if g("TCC_HIT") is not None and g("TCC_MISS") is not None:
        print("- L2 hit rate = %.3f" % (g("TCC_HIT") / (g("TCC_HIT") + g("TCC_MISS"))))
    if g("TCP_TOTAL_CACHE_ACCESSES") and g("TCP_TCC_READ_REQ") is not None:
        print("- L1: %.4g accesses, %.4g read requests to L2 (ratio %.3f)" % (
            g("TCP_TOTAL_CACHE_ACCESSES"), g("TCP_TCC_READ_REQ"), g("TCP_TCC_READ_REQ") / g("TCP_TOTAL_CACHE_ACCESSES")))
    if g("SQ_WAVE_CYCLES"):
        wc = g("SQ_WAVE_CYCLES")
        print("- wave-cycles: WAIT_ANY %.1f%%, WAIT_INST_ANY %.1f%% of SQ_WAVE_CYCLES" % (
            100 * (g("SQ_WAIT_ANY") or 0) / wc, 100 * (g("SQ_WAIT_INST_ANY") or 0) / wc))
    if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CYCLES"):
        print("- SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES = %.3f" % (g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES")))
    if g("GRBM_GUI_ACTIVE"):
        print("- GRBM_GUI_ACTIVE per launch = %.4g (sum over 8 XCDs)" % g("GRBM_GUI_ACTIVE"))
        if k in kt_ms:
            print("- effective shader clock = GRBM_GUI_ACTIVE / 8 / kernel time = %.0f MHz (kernel %.3f ms)" % (
                g("GRBM_GUI_ACTIVE") / 8 / (kt_ms[k] * 1e-3) / 1e6, kt_ms[k]))
    print()

if traffic:
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src_sha = {}
    for f in sorted(glob.glob(os.path.join(root, "ngsdist_amd", "csrc", "accum_*.hip")) +
                    glob.glob(os.path.join(root, "ngsdist_amd", "csrc", "contract_mfma.hip"))):
        src_sha[os.path.basename(f)] = hashlib.sha256(open(f, "rb").read()).hexdigest()[:16]
    with open(os.path.join(out, "traffic.json"), "w") as fh:
        json.dump({"source": os.path.basename(out), "kernel_source_sha16": src_sha, "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; "
                   "FETCH_SIZE x 1024 x 2 (gfx950: reads of a wide coalesced stream are counted at half), WRITE_SIZE x 1024",
                   "per_launch": traffic}, fh, indent=1)
