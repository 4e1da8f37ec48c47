#!/usr/bin/env python3
"""tools/shape_sweep.py n_ind n_sites [kernel] [key=value ...] -- accumulate/reduce time of one matrix for a problem
shape; key=value pairs are launch-geometry fields of ngd_config (n_slices, wg_target, exact_shapes, variant)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ngsdist_amd as N

n_ind, n_sites = int(sys.argv[1]), int(sys.argv[2])
kernel = sys.argv[3] if len(sys.argv) > 3 and "=" not in sys.argv[3] else "mfma"
geom = {k: int(v) for k, v in (a.split("=") for a in sys.argv[3:] if "=" in a)}
e = N.Engine(n_ind, n_sites, indep_geno=kernel in ("mfma", "stream"), kernel=kernel, **geom)
e.synth_fill(5, 0.0)
e.run()
ts = []
for _ in range(5):
    e.run()
    t = e.timing()
    ts.append((t["ms_accum"], t["ms_reduce"], t["ms_total"]))
a, r, tot = (float(np.mean([x[k] for x in ts])) for k in range(3))
ps = N.n_pairs(n_ind) * n_sites
print(("%s " % geom if geom else "") + "%5d x %8d %-8s accum %8.3f ms reduce %6.3f ms total %8.3f ms -> %.3g pair-sites/s, %5.1f TF algorithmic (%.3f of FP64 peak)"
      % (n_ind, n_sites, kernel, a, r, tot, ps / (tot * 1e-3), 6 * ps / a / 1e9, 6 * ps / a / 1e9 / 78.6))
