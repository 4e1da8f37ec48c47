#!/usr/bin/env python3
"""tools/em_shapes.py [n_sites] -- times the EM kernels on the cfg 4 shape (1000 individuals, synthetic GLs):
em_fast (one lane per pair) and the table-driven kernel in each workgroup shape (ngd_config.variant), checks that
they agree on every pair, prints ms per launch."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngsdist_amd as N  # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
shapes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3]
n_ind = int(os.environ.get("N_IND", "1000"))
ref = None
for name, shape in [("em_fast", None)] + [("em_table", k) for k in shapes]:
    with N.Engine(n_ind, n_sites, indep_geno=False, kernel=name, variant=shape or 0) as e:
        e.synth_fill(3)
        ms = []
        for it in range(3):
            s, c = e.run()
            ms.append(e.timing()["ms_accum"])
    if ref is None:
        ref = s
    err = float(np.max(np.abs(s - ref) / np.abs(ref)))
    print("%-9s shape %-4s ms_accum %s  pair-sites/s %.3g  max rel diff vs em_fast %.2e"
          % (name, shape, ["%.2f" % m for m in ms], N.n_pairs(n_ind) * n_sites / (min(ms) * 1e-3), err), flush=True)
