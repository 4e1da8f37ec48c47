#!/usr/bin/env python3
"""tools/em_ab.py [n_sites] [variants...] -- the table-driven EM kernel's workgroup shapes / later-round forms side by side on
the cfg 4 shape (1000 individuals): ms per launch, and whether every sum carries the bits of variant 0's."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngsdist_amd as N  # noqa: E402

# --ref file.npz: the first library to run writes its sums there, a later one (another build: NGSDIST_AMD_LIB) is compared
ref_path = None
if "--ref" in sys.argv:
    k = sys.argv.index("--ref")
    ref_path = sys.argv[k + 1]
    del sys.argv[k:k + 2]
n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
variants = [int(v) for v in sys.argv[2:]] or [4, 0]
saved = dict(np.load(ref_path)) if ref_path and os.path.exists(ref_path) else None
to_save = {}
n_ind = 1000
ref = {}
for case, kw, fill in (("plain", {}, 0.0), ("pairwise_del, 10% missing", {"pairwise_del": True}, 0.1)):
    for v in variants:
        with N.Engine(n_ind, n_sites, indep_geno=False, kernel="em_table", variant=v, **kw) as e:
            e.synth_fill(3, fill)
            ms = []
            for _ in range(4):
                s, c = e.run()
                ms.append(e.timing()["ms_accum"])
            bm = N.Taus(7).block_map(n_sites // 10)
            sb, cb = e.run(bm, 10)  # a bootstrap replicate: the weighted kernel
            r = e.em_work()
        key = case
        if key not in ref:
            ref[key] = (s, c, sb, cb)
        same = all(np.array_equal(a, b) for a, b in zip(ref[key], (s, c, sb, cb)))
        rel = max(float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) for a, b in ((s, ref[key][0]), (sb, ref[key][2])))
        print("%-28s variant %d: %.2f ms per launch (min of %s), rounds per (tile, site) %.3f, vs variant %d: bits equal %s, "
              "max rel diff %.2e, counts equal %s" % (case, v, min(ms), " ".join("%.1f" % x for x in ms), r[1] / max(1, r[0]),
                                                     variants[0], same, rel, np.array_equal(c, ref[key][1])), flush=True)
        tag = "%s/%d" % (case, v)
        if saved is not None:
            print("    against %s: plain pass bits equal %s, bootstrap replicate bits equal %s"
                  % (ref_path, np.array_equal(saved[tag + "/s"], s), np.array_equal(saved[tag + "/sb"], sb)), flush=True)
        else:
            to_save[tag + "/s"], to_save[tag + "/sb"] = s, sb
if ref_path and saved is None:
    np.savez(ref_path, **to_save)
