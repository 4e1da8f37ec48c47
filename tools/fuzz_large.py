#!/usr/bin/env python3
"""tools/fuzz_large.py [first_seed] [n_cases] -- random cases at the sizes where round 6's machinery is live (more than 384 padded
individuals, tens of thousands of sites: one operand image + fix-up pass, device memory mapped in pieces, staged raw uploads
through the ring of pinned buffers, the full-data pass started beside the load), each against a TWO-image engine fed the plain
way: valid-site counts equal, sums within 1e-9 relative -- plain pass, a weighted pass, a bootstrap job (and that job again through
ngd_run_job_dist: the bits of ngd_run_job + ngd_finish) -- and on a handful of pairs against the CPU oracle.  Clusters of nearly identical individuals of random size (a few copies: tile by tile / pair by
pair; hundreds: by one more pass over scratch images).  Prints the failing cases, exit 1 if any."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ngsdist_amd as N
from oracle import oracle as O

RTOL = 1e-9
first, n_cases = (int(sys.argv[1]) if len(sys.argv) > 1 else 1), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)
L = N._lib.load()
bad = 0
t_start = time.time()
for case in range(first, first + n_cases):
    rng = np.random.default_rng(case)
    n_ind = int(rng.choice([385, 400, 449, 512, 530, 640, 700, 900]))
    n_sites = int(rng.choice([8_000, 20_000, 33_333, 60_000]))
    pdel = bool(rng.integers(0, 2))
    miss = float(rng.choice([0.0, 0.05, 0.3])) if pdel else float(rng.choice([0.0, 0.05]))
    em = bool(rng.integers(0, 5) == 0) and n_ind <= 530
    B = int(rng.choice([1, 7, 16, 100, 1000]))
    n_rep = int(rng.choice([0, 2, 5, 33]))
    partials = int(rng.integers(0, 3))
    staged, eager = bool(rng.integers(0, 2)), int(rng.integers(0, 2))
    ring, piece = int(rng.integers(2, 9)), int(rng.choice([1, 4, 32]))
    p = O.synth_indmajor(5000 + case, n_ind, n_sites, miss_frac=miss)
    n_cl = int(rng.choice([0, 3, 12, 150, n_ind // 2])) if not em else 0
    if n_cl:
        eps = float(rng.choice([1e-8, 1e-11, 1e-20]))
        gcl = rng.integers(0, 3, size=n_sites)
        pc = eps * (1 + rng.random((n_cl, n_sites, 3)))
        pc[:, np.arange(n_sites), gcl] = 0
        pc[:, np.arange(n_sites), gcl] = 1 - pc.sum(axis=2)
        who = rng.choice(n_ind, size=n_cl, replace=False)
        p[who] = pc
    tag = (case, n_ind, n_sites, "EM" if em else "indep", pdel, miss, B, n_rep, partials, "staged" if staged else "plain", eager, ring, piece, n_cl)
    n_eff = n_sites - n_sites % B
    t = N.Taus(case)
    maps = np.stack([t.block_map(n_eff // B) for _ in range(n_rep)]) if n_rep else None
    try:
        out = []
        one_call_ok = True
        for which in (0, 1):  # 0: the engine under test; 1: the reference engine (two images / the same EM kernel), plain upload
            kw = dict(single_image=3) if (which and not em) else {}
            with N.Engine(n_ind, n_sites, pairwise_del=pdel, indep_geno=not em, **kw) as e:
                e.set_option("boot_partials", partials)
                if which == 0 and staged:
                    e.set_option("stage_piece_mib", piece).set_option("stage_ring", ring).set_option("eager_full", eager)
                    e.upload_raw_sites(np.ascontiguousarray(p.transpose(1, 0, 2)), 0).commit()
                elif em and staged:
                    # the reference engine of an EM case takes the raw values through K0 too (default ring): a site within ulps
                    # of the EM's stopping threshold lands on one of two adjacent iterates, 0.004 apart in its term, and an ulp
                    # of difference in the prepared input decides which (case 70532: one site of one pair of 74 000; the oracle
                    # and the reference's own em2 flip the same way between the two inputs)
                    e.upload_raw_sites(np.ascontiguousarray(p.transpose(1, 0, 2)), 0).commit()
                else:
                    e.upload_ind_major(p).commit()
                s0, c0 = e.run()
                f0 = e.fixup()
                res = [(s0, c0)]
                if n_eff >= B and n_eff // B:
                    m1 = N.Taus(case + 1).block_map(n_eff // B)
                    res.append(e.run(m1, B))
                if n_rep:
                    S, Cn = e.run_job(maps, B)
                    res += [(S[k], Cn[k]) for k in sorted({0, n_rep})]
                    if which == 0:  # the job and its tail in one call: the bits of ngd_run_job + ngd_finish
                        with np.errstate(all="ignore"):
                            want = N.finish(S.reshape(-1), Cn.reshape(-1), 0, 1 + case % 2).reshape(S.shape)
                            got = e.run_job_dist(maps, B, 1 + case % 2)
                            one_call_ok = np.array_equal(got.view(np.uint64), want.view(np.uint64))
                            if partials == 1 and not one_call_ok:
                                # boot_partials = 1 lets the engine serve the first calls of a geometry WITHOUT the per-block
                                # partial results while their slab would cost more to allocate than it has saved so far
                                # (engine.hip partials_impl, "rent"): two calls of the same job may then take different plans
                                # and agree to rounding, not bit for bit -- ngd_run_job twice does the same
                                fin = np.isfinite(want)
                                one_call_ok = np.array_equal(np.isfinite(got), fin) and bool(
                                    np.all(np.abs(got[fin] - want[fin]) <= RTOL * np.maximum(np.abs(want[fin]), 1e-300)))
                            if not one_call_ok and os.environ.get("NGD_FUZZ_DETAIL"):
                                w = np.argwhere(got.view(np.uint64) != want.view(np.uint64))
                                print("  one call: %d cells differ; first:" % len(w), w[:6].tolist(), [(got[tuple(x)], want[tuple(x)]) for x in w[:6]],
                                      "matrices:", sorted(set(w[:, 0].tolist()))[:12], flush=True)
                                S2, C2 = e.run_job(maps, B)
                                print("  a second ngd_run_job gives the first one's bits:", np.array_equal(S2.view(np.uint64), S.view(np.uint64)),
                                      np.array_equal(C2, Cn), flush=True)
                out.append((res, f0))
        ok = one_call_ok
        for k_res, ((a, ca), (b, cb)) in enumerate(zip(out[0][0], out[1][0])):
            fin = np.isfinite(b) & (cb > 0)
            ok_k = np.array_equal(ca, cb) and np.array_equal(np.isfinite(a), np.isfinite(b))
            if fin.any():
                ok_k = ok_k and bool(np.all(np.abs(a[fin] - b[fin]) <= RTOL * np.maximum(np.abs(b[fin]), 1e-300)))
            if not ok_k and os.environ.get("NGD_FUZZ_DETAIL"):
                rel = np.where(fin, np.abs(a - b) / np.maximum(np.abs(b), 1e-300), 0.0)
                w = np.argsort(rel)[::-1][:4]
                print("  result %d against the reference engine: counts equal %s, worst pairs %s rel %s got %s want %s cnt %s" % (
                    k_res, np.array_equal(ca, cb), w.tolist(), rel[w].tolist(), a[w].tolist(), b[w].tolist(), cb[w].tolist()), flush=True)
            ok = ok and ok_k
        if out[0][1]["skipped"]:
            ok = False
        # a handful of pairs against the oracle (clones among them)
        sub = np.unique(np.concatenate([rng.choice(n_ind, size=6, replace=False), (who[:4] if n_cl else np.array([0], dtype=np.int64))]))
        so, co = O.all_pairs(np.ascontiguousarray(p[sub]), pairwise_del=pdel, indep_geno=not em, n_threads=16)
        idx = np.array([L.ngd_pair_index(n_ind, int(min(x, y)), int(max(x, y))) for k, x in enumerate(sub) for y in sub[k + 1:]])
        s0, c0 = out[0][0][0]
        fin = co > 0
        ok_o = np.array_equal(c0[idx], co) and bool(np.all(np.abs(s0[idx][fin] - so[fin]) <= RTOL * np.maximum(np.abs(so[fin]), 1e-300)))
        if not ok_o and os.environ.get("NGD_FUZZ_DETAIL"):
            print("  against the oracle: counts equal %s, rel %s" % (np.array_equal(c0[idx], co), (np.abs(s0[idx][fin] - so[fin]) / np.abs(so[fin])).tolist()), flush=True)
        ok = ok and ok_o
        if not ok:
            bad += 1
            print("MISMATCH", tag, out[0][1], "" if one_call_ok else "(ngd_run_job_dist against ngd_run_job + ngd_finish)", flush=True)
    except Exception as exc:  # noqa: BLE001
        bad += 1
        print("ERROR", tag, repr(exc), flush=True)
    if (case - first + 1) % 10 == 0:
        print("... %d cases, %d bad, %.0f s" % (case - first + 1, bad, time.time() - t_start), flush=True)
print("fuzz_large: %d cases from seed %d, %d bad" % (n_cases, first, bad))
sys.exit(1 if bad else 0)
