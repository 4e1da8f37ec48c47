#!/usr/bin/env python3
"""tools/fixup_partials_cost.py [n_ind n_sites block n_rep] -- what the fix-up pass of a bootstrap job by per-block partial results costs on
a data set of copies of one individual (every pair noted): tile by tile, or the whole slab once more in the two-operand arithmetic
(engine.hip fixup_partials_by_pass); both routes forced through the test hook, then the engine's own choice; worst relative difference
from a two-image engine on the same data."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NGD_ENABLE_TEST_HOOKS"] = "1"
import ngsdist_amd as N  # noqa: E402

n_ind, n_sites, B, n_rep = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (1000, 100_000, 1000, 4)
rng = np.random.default_rng(1)
g = rng.integers(0, 3, size=n_sites)
p = 1e-9 * (1 + rng.random((n_sites, n_ind, 3)))  # site-major
p[np.arange(n_sites), :, g] = 0
p[np.arange(n_sites), :, g] = 1 - p.sum(axis=2)
maps = np.stack([N.Taus(r + 1).block_map(n_sites // B) for r in range(n_rep)])
with N.Engine(n_ind, n_sites, kernel="mfma", single_image=3) as e2:
    e2.set_option("boot_partials", 2)
    S2, _ = e2.upload_sites(p, 0).commit().run_job(maps, B)
for route in ("tiles", "pass", None):
    if route:
        os.environ["NGD_TEST_FIX_PARTIALS"] = route
    else:
        os.environ.pop("NGD_TEST_FIX_PARTIALS", None)
    with N.Engine(n_ind, n_sites, kernel="mfma", single_image=2) as e:
        e.set_option("boot_partials", 2)
        S, _ = e.upload_sites(p, 0).commit().run_job(maps, B)
        f = e.fixup()
    print("%d individuals x %d sites, %d blocks of %d, %d replicates, route %s: %d pairs noted, %d recomputed, %d skipped, by_pass %d, fix-up %.1f ms; "
          "worst relative difference from the two-image engine %.2e" % (n_ind, n_sites, n_sites // B, B, n_rep, route or "the engine's own choice",
                                                                       f["flagged"], f["recomputed"], f["skipped"], f["by_pass"], f["ms"],
                                                                       float(np.max(np.abs(S - S2) / np.abs(S2)))), flush=True)
