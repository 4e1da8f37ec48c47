#!/usr/bin/env python3
"""tools/fixup_cost.py [n_ind n_sites] -- what the fix-up pass of a one-image engine costs at the edge of its budget
(ngd_internal.h NGD_FIX_WORK = 4.1e9 pair-sites): a data set of copies of one individual, every pair noted and recomputed
with the two-operand arithmetic (fixup.hip); prints the pass's device time and its rate."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngsdist_amd as N  # noqa: E402

n_ind, n_sites = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 800_000)
rng = np.random.default_rng(1)
g = rng.integers(0, 3, size=n_sites)
p = 1e-9 * (1 + rng.random((n_ind, n_sites, 3)))
p[:, np.arange(n_sites), g] = 0
p[:, np.arange(n_sites), g] = 1 - p.sum(axis=2)
with N.Engine(n_ind, n_sites, kernel="mfma", single_image=2) as e:
    e.upload_ind_major(p).commit()
    for _ in range(2):
        t0 = time.perf_counter()
        s, c = e.run()
        dt = time.perf_counter() - t0
        f = e.fixup()
        tm = e.timing()
    ps = f["recomputed"] * n_sites
    print("%d individuals x %d sites: %d pairs flagged, %d recomputed, %d skipped; fix-up %.1f ms (%.3g pair-sites: %.3g pair-sites/s, "
          "%.0f GB/s at 48 B of likelihoods per pair-site); the MFMA pass itself %.2f ms; the call %.1f ms"
          % (n_ind, n_sites, f["flagged"], f["recomputed"], f["skipped"], f["ms"], ps, ps / (f["ms"] * 1e-3 + 1e-12),
             48 * ps / (f["ms"] * 1e-3 + 1e-12) / 1e9, tm["ms_accum"], dt * 1e3))
