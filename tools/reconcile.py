#!/usr/bin/env python3
"""tools/reconcile.py plain.json kt.json -- does the bench line taken under `rocprofv3 --kernel-trace --stats` agree
with an unprofiled line of the same command taken just before it (same box, same lease)?  Prints both; exit 1 if
ms_per_step differs by more than 2 % (tools/profile.sh then takes the pair once more)."""
import json
import sys


def last_line(path):
    return json.loads([ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1])


a, b = last_line(sys.argv[1]), last_line(sys.argv[2])
ra, rb = a["roofline"], b["roofline"]
off = abs(b["ms_per_step"] / a["ms_per_step"] - 1.0)
for name, d, r in (("unprofiled", a, ra), ("kernel-trace pass", b, rb)):
    print("%-18s ms_per_step %.4f, kernel ms per launch mean %.4f min %.4f median %.4f, frac %s, shader clock in the kernel %s MHz"
          % (name, d["ms_per_step"], r["ms_per_launch"], r.get("ms_per_launch_min", float("nan")),
             r.get("ms_per_launch_median", float("nan")), "%.4f" % r["frac"] if r.get("frac") is not None else "n/a",
             "%.0f" % r["shader_clock_mhz"] if r.get("shader_clock_mhz") else "n/a"))
print("ms_per_step of the traced run is %.2f %% off the unprofiled one: %s" % (100 * off, "ok (<= 2 %)" if off <= 0.02 else "NOT within 2 %"))
sys.exit(0 if off <= 0.02 else 1)
