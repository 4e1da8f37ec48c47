#!/usr/bin/env python3
"""tools/big_n_ind.py [n_ind] [kernels...] -- more individuals than the graded suite's 16 000 (and than round 2's limit of
60 000): pairs from the first, middle and last tiles against the oracle.  The MFMA kernel keeps at least 8 slab planes
of n_pad^2 doubles, so it runs out of device memory first (NGD_E_NOMEM, reported); the streaming and EM kernels go on."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngsdist_amd as N  # noqa: E402
from oracle import oracle as O  # noqa: E402

n_ind = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
kernels = sys.argv[2:] or ["em_table", "stream", "mfma", "auto"]
n_sites = 64
idx = [0, 1, 63, 64, 127, 128, n_ind // 2, n_ind - 129, n_ind - 2, n_ind - 1]
sub = np.concatenate([O.synth_indmajor(7, n_ind, n_sites, miss_frac=0.05, i0=i, n_sub=1) for i in idx])
for kernel in kernels:
    indep = kernel in ("mfma", "stream", "auto")
    t0 = time.time()
    try:
        with N.Engine(n_ind, n_sites, indep_geno=indep, kernel=kernel, pairwise_del=True) as e:
            e.synth_fill(7, 0.05)
            gb = e.device_bytes() / 1e9
            s, c = e.run()
            ms = e.timing()["ms_accum"]
    except N.engine.NgdError as exc:
        print("%s, %d individuals: %s" % (kernel, n_ind, exc), flush=True)
        continue
    so, co = O.all_pairs(sub, pairwise_del=True, indep_geno=indep)
    k, worst, ok = 0, 0.0, True
    for a in range(len(idx)):
        for b in range(a + 1, len(idx)):
            g = N.n_pairs(n_ind) - N.n_pairs(n_ind - idx[a]) + (idx[b] - idx[a] - 1)
            ok = ok and c[g] == co[k]
            worst = max(worst, abs(s[g] - so[k]) / abs(so[k]))
            k += 1
    print("%s, %d individuals x %d sites (%.3g pairs, %.0f GB on the device): kernel %.0f ms, %d pairs checked, counts equal %s, "
          "max rel err %.2e, %.0f s in all" % (kernel, n_ind, n_sites, N.n_pairs(n_ind), gb, ms, k, ok, worst, time.time() - t0), flush=True)
    del s, c
