#!/usr/bin/env python3
"""tools/fixup_cost.py [n_ind n_sites] -- what the fix-up pass of a one-image engine costs on a data set of copies of one
individual: every pair noted and recomputed with the two-operand arithmetic (fixup.hip tile by tile, or -- where the tiles
would cost more -- engine.hip fixup_by_pass: the whole matrix once more over scratch images); prints the pass's device time,
which way it went, and the worst relative difference from a two-image engine on the same data."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngsdist_amd as N  # noqa: E402

n_ind, n_sites = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 800_000)
chunk = max(1, min(n_sites, 200_000_000 // (n_ind * 3)))
with N.Engine(n_ind, n_sites, kernel="mfma", single_image=2) as e, N.Engine(n_ind, n_sites, kernel="mfma", single_image=3) as e2:
    rng = np.random.default_rng(1)
    for s0 in range(0, n_sites, chunk):
        n = min(chunk, n_sites - s0)
        g = rng.integers(0, 3, size=n)
        p = 1e-9 * (1 + rng.random((n, n_ind, 3)))  # site-major
        p[np.arange(n), :, g] = 0
        p[np.arange(n), :, g] = 1 - p.sum(axis=2)
        e.upload_sites(p, s0)
        e2.upload_sites(p, s0)
    e.commit()
    e2.commit()
    s2, _ = e2.run()
    for _ in range(2):
        t0 = time.perf_counter()
        s, c = e.run()
        dt = time.perf_counter() - t0
        f = e.fixup()
        tm = e.timing()
    ps = f["recomputed"] * n_sites
    print("%d individuals x %d sites: %d pairs flagged, %d recomputed, %d skipped, %s; fix-up %.1f ms (%.3g pair-sites: %.3g pair-sites/s); "
          "the MFMA pass itself %.2f ms; the call %.1f ms; worst relative difference from the two-image engine %.2e"
          % (n_ind, n_sites, f["flagged"], f["recomputed"], f["skipped"],
             "by one more pass over scratch images" if f["by_pass"] else "tile by tile / pair by pair", f["ms"], ps,
             ps / (f["ms"] * 1e-3 + 1e-12), tm["ms_accum"], dt * 1e3, float(np.max(np.abs(s - s2) / np.abs(s2)))))
