#!/usr/bin/env python3
"""tools/shard_sweep.py n_ind n_sites -- per-rank accumulate time when the pair tiles are dealt over
1, 2, 4, 8 ranks (each rank's share run alone on this GPU): what strong scaling can be expected from
the kernels themselves, before the collective."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ngsdist_amd as N

n_ind, n_sites = int(sys.argv[1]), int(sys.argv[2])
base = None
for world in (1, 2, 4, 8):
    worst = 0.0
    for rank in range(world):
        e = N.Engine(n_ind, n_sites, kernel="mfma", shard_rank=rank, shard_world=world)
        e.synth_fill(3, 0.0)
        e.run()
        ts = []
        for _ in range(3):
            e.run()
            ts.append(e.timing()["ms_total"])
        worst = max(worst, float(np.mean(ts)))
        e.close()
    base = base or worst
    print("world %d: slowest rank %.3f ms per matrix -> speed-up %.2f (efficiency %.2f)" % (world, worst, base / worst, base / worst / world))
