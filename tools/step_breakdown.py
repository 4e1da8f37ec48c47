#!/usr/bin/env python3
"""tools/step_breakdown.py [n_ind n_sites] -- where one small job's latency goes on the host side (cfg 2 by default):
wall time of the engine call against the engine's own event bracket, the timing read-back, gen_dist()'s tail."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngsdist_amd as N  # noqa: E402

n_ind, n_sites = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 100_000)
n_pairs = N.n_pairs(n_ind)
with N.Engine(n_ind, n_sites, kernel="mfma") as e:
    e.synth_fill(3)
    hs = torch.empty(n_pairs, dtype=torch.float64).pin_memory()
    hc = torch.empty(n_pairs, dtype=torch.int64).pin_memory()
    ds, dc = hs.data_ptr(), hc.data_ptr()  # pinned memory is mapped into the device's address space: no copy-out
    cnt = np.full(n_pairs, n_sites, dtype=np.uint64)
    out = np.empty(n_pairs)
    rows = []
    for it in range(300):
        t0 = time.perf_counter()
        e.drop_caches()
        t1 = time.perf_counter()
        e.run_device(ds, dc)
        t2 = time.perf_counter()
        tm = e.timing()
        t3 = time.perf_counter()
        N.finish(hs.numpy(), cnt, 0, 1, out=out)
        t4 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, tm["ms_total"] * 1e-3, tm["ms_accum"] * 1e-3, t3 - t2, t4 - t3, t4 - t0))
    r = np.median(np.array(rows[50:]), axis=0) * 1e6
    print("%d x %d, medians in us: drop_caches %.1f, engine call %.1f (its own event bracket %.1f, accumulation kernel %.1f), "
          "timing() %.1f, ngd_finish %.1f, step %.1f" % (n_ind, n_sites, *r))
