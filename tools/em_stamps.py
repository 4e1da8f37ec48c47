#!/usr/bin/env python3
"""tools/em_stamps.py [n_sites] [shape] -- reads the cycle stamps of a -DNGD_EMT_STAMPS build (tools/em_stamps.sh) of the table-driven
EM kernel (diagnostic build: the 'sums' it returns are per-wavefront cycle totals per phase, not distances)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngsdist_amd as N  # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
shape = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_ind = 1000
nw = 8 if shape in (0, 2, 4) else 4
rpw = 64 // nw
names = ["loop + wait GL loads", "barrier 2 later rounds", "site set-up", "build", "barrier 1", "scan round 1", "barrier 2 round 1", "scan later rounds"]
with N.Engine(n_ind, n_sites, indep_geno=False, kernel="em_table", variant=shape) as e:
    e.synth_fill(3)
    s, c = e.run()
    ms = e.timing()["ms_accum"]
idx = lambda i, j: N.n_pairs(n_ind) - N.n_pairs(n_ind - i) + (j - i - 1)
print("shape %d, %d sites, kernel %.2f ms; tile (0,1), lane 0 of each wavefront; cycles summed over the tile's slices" % (shape, n_sites, ms))
tot = np.zeros(8)
for w in range(nw):
    v = np.array([s[idx(w * rpw + r, 64)] if r < rpw else 0.0 for r in range(8)])
    tot += v
    print("wave %d: " % w + "  ".join("%s %.3g" % (names[k], v[k]) for k in range(8)) + "  | sum %.3g" % v.sum())
print("share:  " + "  ".join("%s %.1f%%" % (names[k], 100 * tot[k] / tot.sum()) for k in range(8)))
