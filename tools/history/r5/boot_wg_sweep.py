#!/usr/bin/env python3
"""tools/boot_wg_sweep.py -- cfg 5 (500 x 5e5, 64 replicates of 1000-site blocks + the full-data matrix): the per-block
partials pass at different workgroup targets (NGD_OPT_BOOT_WG), partials recomputed in every call."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngsdist_amd as N  # noqa: E402
import torch  # noqa: E402

n_ind, n_sites, B, R = 500, 500_000, 1000, 64
rng = N.Taus(12345)
maps = np.stack([rng.block_map(n_sites // B) for _ in range(R)])
mult = np.stack([np.ones(n_sites // B, dtype=np.uint32)] + [np.bincount(m.astype(np.int64), minlength=n_sites // B).astype(np.uint32) for m in maps])
n_pairs = N.n_pairs(n_ind)
d_s = torch.zeros((R + 1, n_pairs), dtype=torch.float64, device="cuda")
d_c = torch.zeros((R + 1, n_pairs), dtype=torch.int64, device="cuda")
for wg in [int(x) for x in sys.argv[1:]] or [1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384]:
    with N.Engine(n_ind, n_sites, kernel="mfma") as e:
        e.synth_fill(5).set_option("boot_wg", wg)
        acc, tot = [], []
        for it in range(12):
            e.drop_caches()
            e.run_batch(mult=mult, block_size=B, d_sum_ptr=d_s.data_ptr(), d_cnt_ptr=d_c.data_ptr())
            t = e.timing()
            if it >= 3:
                acc.append(t["ms_accum"]); tot.append(t["ms_total"])
    print("boot_wg %6d: pass %.3f ms (min %.3f), engine total %.3f ms" % (wg, np.mean(acc), np.min(acc), np.mean(tot)), flush=True)
