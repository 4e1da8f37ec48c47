#!/usr/bin/env python3
"""tools/ab_lines.py <file.json>... -- one line per bench.py JSON line: step, kernel time, clock, frac at that clock"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as exc:
        print(f, "unreadable:", exc)
        continue
    r = d["roofline"]
    mhz = r.get("shader_clock_mhz") or 0
    ms = r.get("ms_per_launch") or 0
    print("%-40s step %8.3f ms  kernel %8.3f ms (min %s)  clock %6.0f MHz  kernel x clock/2400 %8.3f  frac %s  valid %s" % (
        f.split("/")[-1], d["ms_per_step"], ms, r.get("ms_per_launch_min"), mhz, ms * mhz / 2400.0, r.get("frac"), d.get("valid")))
