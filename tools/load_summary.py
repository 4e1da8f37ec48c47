#!/usr/bin/env python3
"""tools/load_summary.py <load_trace dir> [e2e.jsonl] -- condenses a tools/r6_load_trace.sh output directory (rocprofv3 kernel
trace + memory-copy trace of the C++ host on cfg 3, FETCH_SIZE / WRITE_SIZE passes) into the markdown summary kept under
profiles/: what K0 `k_prep_layout` and the host-to-device copies of the load phase take, how busy the copy engine is, and
K0's bytes against its algorithmic 24 B in + 32 B out per (individual, site)."""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
e2e = sys.argv[2] if len(sys.argv) > 2 else None


def one(pattern):
    f = glob.glob(os.path.join(d, pattern))
    return f[0] if f else None


def phases(log):
    for ln in open(log):
        if ln.startswith("> phases"):
            return {k: float(v) for k, v in (t.split("=") for t in ln.split(":", 1)[1].split())}
    return {}


out = []
out.append("# Load phase of the C++ host on cfg 3 (1000 x 1e6, 24 GB file in host memory) under rocprofv3\n")
out.append("`tools/r6_load_trace.sh`: an unprofiled run, `--kernel-trace --memory-copy-trace --stats`, then `--pmc FETCH_SIZE` and "
           "`--pmc WRITE_SIZE` in passes of their own (with `--kernel-trace` only).\n")
pp, pk = phases(os.path.join(d, "plain.log")), phases(os.path.join(d, "kt.log"))
out.append("| run | load [s] | file GB/s | waiting for a buffer | filling buffers | submitting | whole run since main() |")
out.append("|---|---|---|---|---|---|---|")
for name, p in (("unprofiled", pp), ("kernel + copy trace", pk)):
    if p:
        out.append("| %s | %.3f | %.1f | %.3f | %.3f | %.3f | %.3f |" % (name, p["load"], 24.0 / p["load"], p["of_load_wait_buffer"],
                                                                       p["of_load_read"], p["of_load_submit"], p["total_since_main"]))
out.append("")
kt = one("kt/*/*_kernel_trace.csv")
mc = one("kt/*/*_memory_copy_trace.csv")
k0 = [r for r in csv.DictReader(open(kt)) if "k_prep_layout" in r["Kernel_Name"]]
k0_ns = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in k0]
cp = [r for r in csv.DictReader(open(mc)) if r["Direction"] == "MEMORY_COPY_HOST_TO_DEVICE"]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in cp)
big = [x for x in iv if x[1] - x[0] > 100_000]  # the 32-MiB pieces (not the job lists of ngd_create)
t0, t1 = big[0][0], max(e for _, e in big)
busy, (cs, ce) = 0, big[0]
for s, e in big[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
n_el = 1000 * 1_000_000
out.append("**Host-to-device copies** (memory-copy trace): %d pieces, mean %.3f ms each (32 MiB: %.1f GB/s while one runs alone); "
           "first to last byte %.1f ms = **%.1f GB/s**; the copy engine busy %.1f ms of it (%.0f %%), i.e. %.1f GB/s while busy.\n"
           % (len(big), sum(e - s for s, e in big) / len(big) / 1e6, 33.554432 / (sum(e - s for s, e in big) / len(big) / 1e6),
              (t1 - t0) / 1e6, 24.0 / ((t1 - t0) / 1e9), busy / 1e6, 100.0 * busy / (t1 - t0), 24.0 / (busy / 1e9)))
out.append("**K0 `k_prep_layout`** (kernel trace): %d launches (one per piece), mean %.1f us, %.1f ms in all = %.0f %% of the copy "
           "span: the preparation rides in the copies' shadow.\n" % (len(k0_ns), sum(k0_ns) / len(k0_ns) / 1e3, sum(k0_ns) / 1e6,
                                                                   100.0 * sum(k0_ns) / (t1 - t0)))
for name, scale, alg, what in (("fetch", 1024 * 2, 24.0, "FETCH_SIZE x 1024 x 2 (gfx950: half of a wide read's bytes are tallied)"),
                               ("write", 1024, 32.0, "WRITE_SIZE x 1024")):
    f = one(name + "/*/*_counter_collection.csv")
    if not f:
        continue
    tot = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_prep_layout" in r["Kernel_Name"])
    out.append("K0 %s: %s = **%.2f GB** over the load = %.2f B per (individual, site); algorithmic %.0f B (%s).\n"
               % ("reads" if name == "fetch" else "writes", what, tot * scale / 1e9, tot * scale / n_el, alg,
                  "the raw doubles of the file, once" if name == "fetch"
                  else "24 B of the one-image engine's operand image + 8 B of min(p0, p2) beside it"))
rd = one("fetch/*/*_counter_collection.csv")
if rd:
    out.append("K0's own roofline: (24 + 32) B x 1e9 / %.1f ms = **%.2f TB/s = %.2f of the 8 TB/s HBM peak** per launch -- launches of "
               "32 MiB (1.4e6 threads, ~30 us) do not fill the chip for long; it does not matter to the run, which is bound by the "
               "host link: the kernel is busy %.0f %% of the load.\n"
               % (sum(k0_ns) / 1e6, 56e9 / (sum(k0_ns) / 1e9) / 1e12, 56e9 / (sum(k0_ns) / 1e9) / 8e12, 100.0 * sum(k0_ns) / (t1 - t0)))
if e2e:
    for ln in open(e2e):
        j = json.loads(ln)
        if "link_roof" in j:
            out.append("**The link's roof in the same lease** (`tools/pcie_peak`): pinned hipMemcpyAsync host to device %.2f GB/s (128 MiB, "
                       "one stream) ... %.2f (best); a kernel pulling pinned memory %.2f; device to host %.2f.\n"
                       % (j["link_roof"]["h2d_128MiB_1stream_GBps"], j["pinned_h2d_best_GBps"], j["link_roof"]["kernel_pull_256wg_GBps"],
                          j["link_roof"]["d2h_1024MiB_GBps"]))
        elif j.get("workload") == "cfg3":
            out.append("**End to end** (`tools/bench_e2e.py`, %d runs %g s apart): wall %s s; load %.3f s = %.1f GB/s = **%.2f of the "
                       "link's roof**; phases of the best run: %s.\n"
                       % (len(j["wall_s_runs"]), j["config"]["seconds_between_runs"], ", ".join("%.3f" % w for w in j["wall_s_runs"]),
                          j["phases_s"]["load"], j["load_GBps"], j["roofline_load"]["frac"],
                          ", ".join("%s %.3f" % (k, v) for k, v in j["phases_s"].items() if not k.startswith("of_"))))
print("\n".join(out))
