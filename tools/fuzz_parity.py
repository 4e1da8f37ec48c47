#!/usr/bin/env python3
"""tools/fuzz_parity.py [first_seed] [n_cases] -- the sweep of tests/test_gpu_parity.py::test_random_shapes_flags_and_plans
with other seeds and more cases: odd shapes, every kernel, random flags / block sizes / replicate counts / plans / launch
geometry, ngd_run_job against the CPU oracle (test infrastructure; needs a GPU).  Prints the failing cases, exit 1 if any."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ngsdist_amd as N
from oracle import oracle as O

RTOL = 1e-9
first, n_cases = (int(sys.argv[1]) if len(sys.argv) > 1 else 1), (int(sys.argv[2]) if len(sys.argv) > 2 else 300)
kernels = ["stream", "mfma", "em_faithful", "em_fast", "em_table"]
bad = 0
t_start = time.time()
for case in range(first, first + n_cases):
    rng = np.random.default_rng(case)
    kernel = kernels[case % len(kernels)]
    indep = kernel in ("stream", "mfma")
    n_ind = int(rng.choice([2, 3, 5, 15, 16, 17, 31, 33, 63, 64, 65, 66, 127, 128, 129, 140, 193, 200]))
    n_sites = int(rng.choice([1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 127, 257, 700, 1000, 1025]))
    if not indep and n_ind > 140:
        n_sites = min(n_sites, 257)
    pdel = bool(rng.integers(0, 2))
    miss = float(rng.choice([0.0, 0.05, 0.3, 0.9]))
    B = min(int(rng.choice([1, 2, 3, 4, 7, 8, 12, 50, 100])), n_sites)
    n_rep = int(rng.choice([0, 1, 2, 3, 5, 17, 33]))
    partials, em_batch = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    em_spill = int(rng.integers(0, 3))  # 0 off, 1 from three matrices on, 2 from two on (table-driven EM kernel)
    spill_bytes = int(rng.choice([0, 0, 1 << 20, 3 << 20])) if em_spill else 0  # a few k-groups of terms per chunk
    geom = dict(n_slices=int(rng.choice([0, 0, 1, 3, 8, 16])), exact_shapes=int(rng.integers(0, 7)) if kernel == "mfma" else 0,  # (+ 7 below)
                variant=int(rng.integers(0, 5)) if kernel == "em_table" else 0)
    # (drawn from a generator of their own: the cases of earlier rounds keep their shapes)
    rng_si = np.random.default_rng(7_000_000 + case)
    single = kernel == "mfma" and bool(rng_si.integers(0, 2))  # ngd_config.single_image; the scratch: 4 GB or the smallest ranges
    single_bytes = int(rng_si.choice([0, 1, 1 << 20])) if single else 0
    if kernel == "mfma" and not single:
        geom["single_image"] = int(rng_si.choice([0, 3]))  # the engine's choice / two images always
    if single:
        geom["single_image"] = int(rng_si.integers(1, 3))  # 1: the second image formed in ranges; 2: congruent coordinates
        if geom["single_image"] == 1:
            geom["second_image_bytes"] = int(rng_si.choice([0, 0, 1 << 20, 2 << 20]))  # part of the second image kept resident
    score = O.score_matrix(bool(rng.integers(0, 2)))
    model = int(rng.integers(0, 3))
    p = O.synth_indmajor(1000 + case, n_ind, n_sites, miss_frac=miss)
    if rng.integers(0, 4) == 0:  # called genotypes: one-hot vectors, sums must be bit-exact
        g = np.argmax(p, axis=-1)
        p = np.eye(3)[g]
    # (round 5, a generator of its own again) exact_shapes = 7 -- triangular blocks on the diagonal of the full-pattern form --
    # and nearly identical individuals: copies of one individual whose likelihoods are confident to eps, the pairs a
    # one-image engine in congruent coordinates recomputes with the two-operand arithmetic (fixup.hip)
    rng_r5 = np.random.default_rng(9_000_000 + case)
    if kernel == "mfma" and rng_r5.integers(0, 4) == 0:
        geom["exact_shapes"] = 7
    if indep and n_ind >= 5 and rng_r5.integers(0, 3) == 0:
        eps = float(rng_r5.choice([1e-7, 1e-9, 1e-13, 1e-22]))
        n_cl = int(rng_r5.integers(2, min(n_ind, 12) + 1))
        # (--avg_nuc_dist: two copies of a heterozygote are half a difference apart, parse_args.cpp:134-137 -- homozygotes only)
        gcl = rng_r5.integers(0, 2, size=n_sites) * 2 if np.ravel(score)[4] != 0 else rng_r5.integers(0, 3, size=n_sites)
        pc = eps * (1 + rng_r5.random((n_cl, n_sites, 3)))
        pc[:, np.arange(n_sites), gcl] = 0
        pc[:, np.arange(n_sites), gcl] = 1 - pc.sum(axis=2)
        who = rng_r5.choice(n_ind, size=n_cl, replace=False)
        p = p.copy()
        p[who] = pc
    n_eff = n_sites - n_sites % B
    t = N.Taus(case)
    maps = np.stack([t.block_map(n_eff // B) for _ in range(n_rep)]) if n_rep else None
    tag = (case, kernel, n_ind, n_sites, pdel, miss, B, n_rep, partials, em_batch, em_spill, spill_bytes, geom, single_bytes)
    big = False
    mode = int(rng.integers(0, 3))  # 0: one engine; 1: site ranges (partial multiplicities); 2: pair-tile shards
    tag = tag + (("one", "site ranges", "pair tiles")[mode],)
    try:
        if mode:
            world = int(rng.integers(2, 5))
            k_eff = n_eff // B if B else 0
            if mode == 1:  # contiguous ranges of whole blocks; the tail (n_sites - n_eff) rides in the last range
                cut_blocks = sorted(int(x) for x in rng.integers(0, k_eff + 1, size=world - 1))
                cuts = [0] + [c * B for c in cut_blocks] + [n_sites]
            mults = None
            if n_rep:
                mults = np.stack([np.bincount(m.astype(np.int64), minlength=k_eff) for m in maps]).astype(np.uint32)
                if n_eff <= 64 and n_ind <= 33 and rng.integers(0, 2):  # multiplicities no block map would give (many weight planes)
                    mults = (mults * rng.integers(1, 3000, size=mults.shape)).astype(np.uint32)
                    big = True
            n_p = N.n_pairs(n_ind)
            S = np.zeros((n_rep + 1, n_p))
            Cn = np.zeros((n_rep + 1, n_p), dtype=np.uint64)
            for r in range(world):
                if mode == 1:
                    lo, hi = cuts[r], cuts[r + 1]
                    if hi == lo:
                        continue
                    kw = {}
                    sub = p[:, lo:hi]
                    b_lo, b_hi = min(lo, n_eff) // B, min(hi, n_eff) // B
                else:
                    lo, hi, sub, b_lo, b_hi = 0, n_sites, p, 0, k_eff
                    kw = dict(shard_rank=r, shard_world=world)
                with N.Engine(n_ind, hi - lo, score=score, pairwise_del=pdel, indep_geno=indep, kernel=kernel, **geom, **kw) as e:
                    e.set_option("boot_partials", partials).set_option("em_batch", em_batch)
                    e.set_option("em_spill", em_spill).set_option("em_spill_bytes", spill_bytes)
                    if single:
                        e.set_option("single_image_bytes", single_bytes)
                    e.upload_ind_major(sub).commit()
                    s0, c0 = e.run()
                    S[0] += s0
                    Cn[0] += c0
                    if n_rep and b_hi > b_lo:
                        sb, cb = e.run_batch(mult=np.ascontiguousarray(mults[:, b_lo:b_hi]), block_size=B)
                        S[1:] += sb
                        Cn[1:] += cb
        else:
          with N.Engine(n_ind, n_sites, score=score, pairwise_del=pdel, indep_geno=indep, kernel=kernel, **geom) as e:
            e.set_option("boot_partials", partials).set_option("em_batch", em_batch)
            e.set_option("em_spill", em_spill).set_option("em_spill_bytes", spill_bytes)
            if single:
                e.set_option("single_image_bytes", single_bytes)
            # (round 6, a generator of its own) a quarter of the one-engine cases arrive as RAW chunks through the staged
            # upload -- a ring of 2..8 pinned buffers of 1 MiB, K0 on the device, device memory of the images in pieces --
            # with the full-data pass started beside the load where the kernel takes slice ranges (NGD_OPT_EAGER_FULL)
            rng_r6 = np.random.default_rng(11_000_000 + case)
            if rng_r6.integers(0, 4) == 0:
                e.set_option("stage_piece_mib", 1).set_option("stage_ring", int(rng_r6.integers(2, 9)))
                e.set_option("eager_full", int(rng_r6.integers(0, 2)))
                e.upload_raw_sites(np.ascontiguousarray(p.transpose(1, 0, 2)), 0).commit()
                tag = tag + ("staged",)
            else:
                e.upload_ind_major(p).commit()
            S, Cn = e.run_job(maps, B)
            # (round 6) a third of the one-engine cases also take the job and its tail in ONE call (ngd_run_job_dist: groups of
            # replicates copied out beside the later groups' reductions, the host finishing chunks as they land): the same bits
            # as ngd_run_job + ngd_finish on every matrix
            if rng_r6.integers(0, 3) == 0:
                with np.errstate(all="ignore"):
                    want = N.finish(S.reshape(-1), Cn.reshape(-1), 0, model).reshape(S.shape)
                    got = e.run_job_dist(maps, B, model)
                if not np.array_equal(got.view(np.uint64), want.view(np.uint64)):
                    bad += 1
                    print("MISMATCH", tag, "ngd_run_job_dist against ngd_run_job + ngd_finish", flush=True)
                tag = tag + ("one_call",)
        for m in sorted({0, n_rep // 2, n_rep}):
            src = None if m == 0 else O.boot_site_src(maps[m - 1], B)
            if m and big:
                src = np.concatenate([np.repeat(np.arange(b * B, (b + 1) * B), int(k)) for b, k in enumerate(mults[m - 1])]).astype(np.uint64)
            so, co = O.all_pairs(p, score=score, pairwise_del=pdel, indep_geno=indep, site_src=src,
                                 n_sites=n_sites if m == 0 else (len(src) if big else n_eff), n_threads=8)
            ok = np.array_equal(Cn[m], co)
            fin = np.isfinite(so)
            ok = ok and np.array_equal(np.isfinite(S[m]), fin)
            if fin.any():
                d = np.abs(S[m][fin] - so[fin])
                ok = ok and bool(np.all(d <= RTOL * np.maximum(np.abs(so[fin]), 1e-300)))
            with np.errstate(all="ignore"):
                ok = ok and np.array_equal(N.finish(S[m], Cn[m], 0, model), O.finish(S[m], Cn[m], 0, model), equal_nan=True)
            if not ok:
                bad += 1
                print("MISMATCH", tag, "matrix", m, flush=True)
                if os.environ.get("NGD_FUZZ_DETAIL"):  # which check, and where
                    w = np.flatnonzero(Cn[m] != co)
                    print("  counts equal:", w.size == 0, w[:5], Cn[m][w[:5]], co[w[:5]])
                    print("  finite pattern equal:", np.array_equal(np.isfinite(S[m]), fin))
                    if fin.any():
                        rel = np.abs(S[m] - so) / np.maximum(np.abs(so), 1e-300)
                        rel[~fin] = 0
                        w = np.argsort(rel)[::-1][:5]
                        print("  worst pairs:", w, "got", S[m][w], "want", so[w], "rel", rel[w], "cnt", co[w], flush=True)
    except Exception as ex:  # noqa: BLE001
        bad += 1
        print("ERROR", tag, repr(ex), flush=True)
    if (case - first) % 50 == 49:
        print("... %d cases, %d bad, %.0f s" % (case - first + 1, bad, time.time() - t_start), flush=True)
print("fuzz: %d cases from seed %d, %d bad" % (n_cases, first, bad))
sys.exit(1 if bad else 0)
