"""BASELINE.json's full sizes on the GPU.  Where the CPU oracle would take hours, parity is shown through
(a) two independent device kernels agreeing on every pair, (b) the oracle on a few pairs over all sites,
(c) size-independent properties of the path (additivity over site blocks, invariance under a
permutation of the block map)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def N():
    import ngsdist_amd
    return ngsdist_amd


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.abs(b)))


def test_cfg2_every_pair_against_the_oracle():
    """configs[1]: n_ind=200, n_sites=1e5, GL, --indep_geno, model 0"""
    n_ind, n_sites = 200, 100_000
    p = O.synth_indmajor(2, n_ind, n_sites)
    so, co = O.all_pairs(p, n_threads=16)
    with N().Engine(n_ind, n_sites, kernel="mfma") as e:
        s, c = e.synth_fill(2).run()
    assert np.array_equal(c, co) and rel(s, so) < RTOL
    d, do = N().finish(s, c, 0, 0), O.finish(so, co, 0, 0)
    assert rel(d, do) < RTOL


@pytest.fixture(scope="module")
def cfg3_mfma():
    """configs[2]: n_ind=1000, n_sites=1e6, GL, --indep_geno, both operand images (51 GB resident; the engine's own choice
    at this size is ONE image + the fix-up pass: test_cfg3_single_image_engine)"""
    e = N().Engine(1000, 1_000_000, kernel="mfma", single_image=3)
    e.synth_fill(3)
    yield e
    e.close()


def test_cfg3_oracle_on_a_few_pairs_over_all_sites(cfg3_mfma):
    s, c = cfg3_mfma.run()
    assert np.all(c == 1_000_000)
    idx = [0, 63, 64, 127, 128, 500, 998, 999]  # block / tile edges included
    sub = np.concatenate([O.synth_indmajor(3, 1000, 1_000_000, i0=i, n_sub=1) for i in idx])
    so, _ = O.all_pairs(sub, n_threads=16)
    k = 0
    for a in range(len(idx)):
        for b in range(a + 1, len(idx)):
            g = s[N().n_pairs(1000) - N().n_pairs(1000 - idx[a]) + (idx[b] - idx[a] - 1)]
            assert abs(g - so[k]) / so[k] < RTOL
            k += 1
    # a distance is a mean of per-site terms in [0, 1]
    d = N().finish(s, c, 0, 0)
    assert d.min() > 0 and d.max() < 1


def test_cfg3_block_additivity_and_permutation_invariance(cfg3_mfma):
    e = cfg3_mfma
    B, nb = 1000, 1000
    full, _ = e.run()
    ident = np.arange(nb, dtype=np.uint64)
    s_id, c_id = e.run(ident, B)
    assert np.all(c_id == 1_000_000) and rel(s_id, full) < 1e-12  # same sites, different summation tree
    rng = np.random.default_rng(0)
    perm = rng.permutation(ident)
    s_perm, _ = e.run(perm, B)
    assert np.array_equal(s_perm, s_id)  # multiplicities are what matter, not the order of the draws
    lo = np.repeat(np.arange(nb // 2, dtype=np.uint64), 2)        # first half of the blocks, twice each
    hi = np.repeat(np.arange(nb // 2, nb, dtype=np.uint64), 2)    # second half, twice each
    s_lo, _ = e.run(lo, B)
    s_hi, _ = e.run(hi, B)
    assert rel(s_lo + s_hi, 2 * full) < 1e-12
    # and the one-pass-per-replicate path agrees with the partial-sum path
    e.set_option("boot_partials", 0)
    try:
        s_w, _ = e.run(perm, B)
    finally:
        e.set_option("boot_partials", 1)
    assert rel(s_w, s_perm) < 1e-12


def test_cfg3_streaming_kernel_agrees_on_every_pair(cfg3_mfma):
    """two independent device implementations (one wavefront per pair vs FP64 MFMA tiles), all 499 500 pairs"""
    full, _ = cfg3_mfma.run()
    with N().Engine(1000, 1_000_000, kernel="stream") as e:
        s, c = e.synth_fill(3).run()
    assert np.all(c == 1_000_000)
    assert rel(s, full) < RTOL


def test_cfg3_single_image_engine(cfg3_mfma):
    """ngd_config.single_image at full size.  1: 30 GB resident instead of 51 (p only; q = score . p formed a range of
    sites at a time), the whole pass equal to rounding, per-block partial sums bit for bit.  2 (the engine's own choice at
    this size): 35 GB (one image in coordinates in which the score matrix is diagonal + min(p0, p2) beside it for the
    fix-up pass of nearly identical pairs -- none in this data set), everything equal to rounding."""
    full, cnt = cfg3_mfma.run()
    perm = np.random.default_rng(0).permutation(np.arange(1000, dtype=np.uint64))
    cfg3_mfma.set_option("boot_partials", 2)
    try:
        s_perm, _ = cfg3_mfma.run(perm, 1000)
    finally:
        cfg3_mfma.set_option("boot_partials", 1)
    with N().Engine(1000, 1_000_000, kernel="mfma", single_image=True) as e:
        e.synth_fill(3)
        assert e.device_bytes() < 31e9  # (both images: 51.3 GB)
        s, c = e.run()
        assert np.array_equal(c, cnt) and rel(s, full) < 1e-12
        assert np.array_equal(e.run()[0], s)
        e.set_option("boot_partials", 2)  # (per-block partial sums from the first replicate on, as the fixture's engine has them by now)
        sp, _ = e.run(perm, 1000)
        assert np.array_equal(sp, s_perm)
        e.set_option("boot_partials", 0)
        sw, _ = e.run(perm, 1000)
        assert rel(sw, s_perm) < 1e-12
    # single_image = 2, the one image in congruent coordinates (what single_image = 0 picks here): 35 GB, sums to 1e-12,
    # partial sums too
    with N().Engine(1000, 1_000_000, kernel="mfma") as e:
        assert e.image_mode() == (2, True)
        e.synth_fill(3)
        assert e.device_bytes() < 36e9
        s, c = e.run()
        assert e.fixup() == {"flagged": 0, "recomputed": 0, "skipped": 0, "ms": 0.0, "by_pass": 0}
        assert np.array_equal(c, cnt) and rel(s, full) < 1e-12
        assert np.array_equal(e.run()[0], s)
        e.set_option("boot_partials", 2)
        sp, _ = e.run(perm, 1000)
        assert rel(sp, s_perm) < 1e-12
        e.set_option("boot_partials", 0)
        sw, _ = e.run(perm, 1000)
        assert rel(sw, s_perm) < 1e-12
    # ... and with 10 GB of the second image kept resident (ngd_config.second_image_mib): the same results
    with N().Engine(1000, 1_000_000, kernel="mfma", single_image=True, second_image_bytes=10 << 30) as e:
        e.synth_fill(3)
        assert 38e9 < e.device_bytes() < 42e9  # (29.7 GB + 10.7; both images: 51.3)
        s, c = e.run()
        assert np.array_equal(c, cnt) and rel(s, full) < 1e-12
        e.set_option("boot_partials", 2)
        sp, _ = e.run(perm, 1000)
        assert np.array_equal(sp, s_perm)


def test_one_image_engine_recomputes_a_large_data_set_of_clones_whatever_it_costs():
    """1400 copies of one individual x 250 000 sites: 979 300 noted pairs in 3916 tiles = 4.2e9 pair-sites of recomputation --
    above the 4.1e9 at which the engines of rounds 4-5 gave up on ALL of them and returned the one-image sums (absolute
    error 4e-17 per site, i.e. 1e-3 relative on these sums of ~1e-8 per site).  The default engine now recomputes every one:
    1e-9 relative against the two-image engine on every pair and against the oracle on the pairs of 12 individuals -- by ONE
    more pass in the two-image arithmetic over scratch images (engine.hip fixup_by_pass), since 3916 tiles cost more."""
    n_ind, n_sites, eps, chunk = 1400, 250_000, 1e-9, 25_000
    sub = np.array([0, 1, 15, 16, 17, 200, 640, 641, 900, 1398, 1399, 777])
    keep = []
    with N().Engine(n_ind, n_sites, kernel="mfma") as e1, N().Engine(n_ind, n_sites, kernel="mfma", single_image=3) as e2:
        assert e1.image_mode() == (2, True) and e2.image_mode()[0] == 3
        rng = np.random.default_rng(11)
        for s0 in range(0, n_sites, chunk):
            g = rng.integers(0, 3, size=chunk)
            p = eps * (1 + rng.random((chunk, n_ind, 3)))  # site-major, as the file has it (read_data.cpp:28-31)
            p[np.arange(chunk), :, g] = 0
            p[np.arange(chunk), :, g] = 1 - p.sum(axis=2)
            e1.upload_sites(p, s0)
            e2.upload_sites(p, s0)
            keep.append(np.ascontiguousarray(p[:, sub, :].transpose(1, 0, 2)))
        s1, c1 = e1.commit().run()
        f = e1.fixup()
        s2, c2 = e2.commit().run()
        # a weighted pass (one bootstrap replicate, the weights inside the accumulation) the same way
        m = N().Taus(3).block_map(n_sites // 1000)
        for e in (e1, e2):
            e.set_option("boot_partials", 0)
        w1, wc1 = e1.run(m, 1000)
        fw = e1.fixup()
        w2, wc2 = e2.run(m, 1000)
    assert fw["by_pass"] == 1 and fw["skipped"] == 0 and np.array_equal(wc1, wc2) and rel(w1, w2) < RTOL
    # (3916 tiles x 250 000 sites would take 0.4 s tile by tile: the whole matrix once more in the two-image arithmetic instead)
    assert f["by_pass"] == 1 and f["ms"] < 150
    assert f["flagged"] == f["recomputed"] == N().n_pairs(n_ind) and f["skipped"] == 0
    assert np.array_equal(c1, c2) and rel(s1, s2) < RTOL
    so, co = O.all_pairs(np.concatenate(keep, axis=1), n_threads=16)
    idx = [N()._lib.load().ngd_pair_index(n_ind, int(min(a, b)), int(max(a, b))) for k, a in enumerate(sub) for b in sub[k + 1:]]
    assert rel(s1[idx], so) < RTOL


def test_cfg4_em_forms_agree_on_every_pair():
    """configs[3] shape (n_ind=1000, EM, JC69) on 20 000 sites, every pair: the table-driven kernel and the per-pair
    fast form vs the form whose iterates are bit-identical to emOptim2.cpp's; plus the oracle on a few pairs."""
    n_ind, n_sites = 1000, 20_000
    res = {}
    for k in ("em_table", "em_fast", "em_faithful"):
        with N().Engine(n_ind, n_sites, indep_geno=False, kernel=k) as e:
            res[k] = e.synth_fill(3).run()
    for k in ("em_table", "em_fast"):
        assert np.array_equal(res[k][1], res["em_faithful"][1])
        assert rel(res[k][0], res["em_faithful"][0]) < RTOL
    assert rel(res["em_table"][0], res["em_fast"][0]) < 1e-12
    idx = [0, 15, 16, 999]
    sub = np.concatenate([O.synth_indmajor(3, n_ind, n_sites, i0=i, n_sub=1) for i in idx])
    so, co = O.all_pairs(sub, indep_geno=False, n_threads=8)
    k = 0
    for a in range(len(idx)):
        for b in range(a + 1, len(idx)):
            g = res["em_fast"][0][N().n_pairs(n_ind) - N().n_pairs(n_ind - idx[a]) + (idx[b] - idx[a] - 1)]
            assert abs(g - so[k]) / so[k] < RTOL
            k += 1
    with np.errstate(all="ignore"):
        d = N().finish(res["em_fast"][0], res["em_fast"][1], 0, 2)
    assert np.isfinite(d).all()


def test_cfg5_bootstrap_replicates_against_the_oracle_on_a_few_pairs():
    """configs[4]: n_ind=500, n_sites=5e5, 64 replicates of 1000-site blocks (first 3 checked)"""
    n_ind, n_sites, B = 500, 500_000, 1000
    rng_g, rng_o = N().Taus(12345), O.Taus(12345)
    idx = [0, 127, 128, 499]
    sub = np.concatenate([O.synth_indmajor(5, n_ind, n_sites, i0=i, n_sub=1) for i in idx])
    with N().Engine(n_ind, n_sites, kernel="mfma") as e:
        e.synth_fill(5)
        for rep in range(3):
            bm = rng_g.block_map(n_sites // B)
            assert np.array_equal(bm, rng_o.block_map(n_sites // B))
            s, c = e.run(bm, B)
            so, co = O.all_pairs(sub, site_src=O.boot_site_src(bm, B), n_threads=8)
            k = 0
            for a in range(len(idx)):
                for b in range(a + 1, len(idx)):
                    pk = N().n_pairs(n_ind) - N().n_pairs(n_ind - idx[a]) + (idx[b] - idx[a] - 1)
                    assert c[pk] == co[k] and abs(s[pk] - so[k]) / so[k] < RTOL
                    k += 1


def test_cfg5_whole_job_in_one_batch_with_pairwise_deletion():
    """configs[4] as ONE ngd_run_batch call: the full-data matrix (all-ones multiplicities) + 64 replicates, with
    5 % missing sites and --pairwise_del so that the per-block count partials carry real information.  Checked:
    the all-ones row against a plain run (counts exact, sums to rounding), the last replicate against the oracle on
    a few pairs over all its sites, and the first replicate against a single ngd_run (identical bits)."""
    n_ind, n_sites, B, R = 500, 500_000, 1000, 64
    n_blocks = n_sites // B
    rng_g = N().Taus(12345)
    maps = np.stack([rng_g.block_map(n_blocks) for _ in range(R)])
    mult = np.concatenate([np.ones((1, n_blocks), dtype=np.uint32),
                           np.stack([np.bincount(m.astype(np.int64), minlength=n_blocks) for m in maps]).astype(np.uint32)])
    idx = [0, 127, 128, 499]
    sub = np.concatenate([O.synth_indmajor(5, n_ind, n_sites, miss_frac=0.05, i0=i, n_sub=1) for i in idx])
    with N().Engine(n_ind, n_sites, kernel="mfma", pairwise_del=True) as e:
        e.synth_fill(5, 0.05)
        S, Cn = e.run_batch(mult=mult, block_size=B)
        s0, c0 = e.run()
        s1, c1 = e.run(maps[0], B)
    assert np.array_equal(Cn[0], c0) and rel(S[0], s0) < 1e-12
    assert np.array_equal(Cn[1], c1) and np.array_equal(S[1], s1)
    so, co = O.all_pairs(sub, pairwise_del=True, site_src=O.boot_site_src(maps[-1], B), n_threads=8)
    k = 0
    for a in range(len(idx)):
        for b in range(a + 1, len(idx)):
            pk = N().n_pairs(n_ind) - N().n_pairs(n_ind - idx[a]) + (idx[b] - idx[a] - 1)
            assert Cn[-1][pk] == co[k] and abs(S[-1][pk] - so[k]) / so[k] < RTOL
            k += 1


def test_many_individuals_em_path_against_the_oracle():
    """n_ind = 2000 on the EM path (32 x 32 tiles of 64, 2.0e6 pairs), few sites: every pair against the oracle."""
    n_ind, n_sites = 2000, 48
    p = O.synth_indmajor(10, n_ind, n_sites, miss_frac=0.1)
    so, co = O.all_pairs(p, pairwise_del=True, indep_geno=False, n_threads=16)
    with N().Engine(n_ind, n_sites, kernel="em_table", pairwise_del=True, indep_geno=False) as e:
        s, c = e.synth_fill(10, 0.1).run()
    assert np.array_equal(c, co) and rel(s, so) < RTOL


def test_many_individuals_against_the_oracle():
    """n_ind = 3000 (24 x 24 pair tiles, 4.5e6 pairs), few sites: every pair against the oracle."""
    n_ind, n_sites = 3000, 200
    p = O.synth_indmajor(9, n_ind, n_sites, miss_frac=0.1)
    so, co = O.all_pairs(p, pairwise_del=True, n_threads=16)
    with N().Engine(n_ind, n_sites, kernel="mfma", pairwise_del=True) as e:
        s, c = e.synth_fill(9, 0.1).run()
    assert np.array_equal(c, co) and rel(s, so) < RTOL


@pytest.fixture(scope="module")
def cfg4_em():
    """configs[3]: n_ind=1000, n_sites=1e6, GL, EM path (no --indep_geno), JC69 -- at its stated size"""
    e = N().Engine(1000, 1_000_000, indep_geno=False, kernel="em_table")
    e.synth_fill(3)
    yield e
    e.close()


def test_cfg4_full_size_oracle_on_a_few_pairs_over_all_sites(cfg4_em):
    """the EM of all 1e6 sites of 6 pairs on the CPU (the oracle's em2 is bit-identical to the reference's own,
    tests/test_oracle_golden.py) against the device's 4.995e11 pair-sites; then the JC69 tail"""
    e = cfg4_em
    s, c = e.run()
    assert np.all(c == 1_000_000)
    idx = [0, 63, 64, 999]  # both sides of a tile edge, first and last individual
    sub = np.concatenate([O.synth_indmajor(3, 1000, 1_000_000, i0=i, n_sub=1) for i in idx])
    so, co = O.all_pairs(sub, indep_geno=False, n_threads=16)
    k = 0
    for a in range(len(idx)):
        for b in range(a + 1, len(idx)):
            g = s[N().n_pairs(1000) - N().n_pairs(1000 - idx[a]) + (idx[b] - idx[a] - 1)]
            assert abs(g - so[k]) / so[k] < RTOL
            k += 1
    assert k == 6
    with np.errstate(all="ignore"):
        d = N().finish(s, c, 0, 2)
    assert np.isfinite(d).all() and d.min() > 0
    tile_sites, rounds = e.em_work()
    assert tile_sites == 136 * 1_000_000 and 1.0 <= rounds / tile_sites <= 4.0


def test_cfg4_full_size_block_additivity(cfg4_em):
    """size-independent properties at full size: the sum over sites is additive over blocks of sites and depends
    only on block multiplicities (the per-block partial sums of the bootstrap path vs the plain pass)"""
    e = cfg4_em
    B, nb = 10_000, 100
    full, _ = e.run()
    ident = np.arange(nb, dtype=np.uint64)
    s_id, c_id = e.run(ident, B)
    assert np.all(c_id == 1_000_000) and rel(s_id, full) < 1e-12
    perm = np.random.default_rng(0).permutation(ident)
    s_perm, _ = e.run(perm, B)
    assert np.array_equal(s_perm, s_id)
    lo = np.repeat(np.arange(nb // 2, dtype=np.uint64), 2)
    hi = np.repeat(np.arange(nb // 2, nb, dtype=np.uint64), 2)
    s_lo, _ = e.run(lo, B)
    s_hi, _ = e.run(hi, B)
    assert rel(s_lo + s_hi, 2 * full) < 1e-12


def test_em_bootstrap_job_of_a_hundred_replicates_by_spilled_terms():
    """cfg 4's shape on 1e5 sites with the reference's own kind of bootstrap (parse_args.cpp:29-31: EM, blocks far too
    small for per-block partials): the full data + 100 replicates of 10-site blocks through ngd_run_job -- ONE EM pass,
    its terms spilled in chunks and contracted with the 101 weight vectors by FP64 MFMA (contract_mfma.hip).  Several
    chunks (2 GB of scratch), counts exact, matrix 0 and four replicates <= 1e-12 from their own passes, the identity
    block map reproduces the full data, and 6 pairs of the last replicate against the oracle over all its sites."""
    n_ind, n_sites, B, n_rep = 1000, 100_000, 10, 100
    rng = N().Taus(11)
    maps = np.stack([rng.block_map(n_sites // B) for _ in range(n_rep)])
    maps[1] = np.arange(n_sites // B, dtype=np.uint64)  # every block once: the full data
    with N().Engine(n_ind, n_sites, indep_geno=False, kernel="em_table") as e:
        e.synth_fill(3).set_option("boot_partials", 0).set_option("em_spill_bytes", 2 << 30)
        S, Cn = e.run_job(maps, B)
        assert e.timing()["launches"] == 1
        s0, c0 = e.run()
        assert np.array_equal(Cn[0], c0) and rel(S[0], s0) < 1e-12 and rel(S[2], s0) < 1e-12
        for r in (0, 49, 98, 99):
            s1, c1 = e.run(maps[r], B)
            assert np.array_equal(Cn[r + 1], c1) and rel(S[r + 1], s1) < 1e-12
    idx = [0, 1, n_ind // 2, n_ind - 1]
    sub = np.concatenate([O.synth_indmajor(3, n_ind, n_sites, i0=i, n_sub=1) for i in idx])
    so, co = O.all_pairs(sub, indep_geno=False, site_src=O.boot_site_src(maps[-1], B), n_sites=n_sites, n_threads=6)
    k = 0
    for a in range(4):
        for b in range(a + 1, 4):
            g = S[-1][N().n_pairs(n_ind) - N().n_pairs(n_ind - idx[a]) + (idx[b] - idx[a] - 1)]
            assert abs(g - so[k]) / abs(so[k]) < RTOL
            k += 1


@pytest.mark.parametrize("kernel,indep", [("mfma", True), ("stream", True), ("em_table", False), ("em_fast", False)])
def test_tens_of_thousands_of_individuals(kernel, indep):
    """n_ind = 16 000 (1.28e8 pairs, a 2 GB slab plane; 20 000 and 40 000 were run by hand at the end of round 2): index
    widths and slab sizes -- pairs from the first, middle and last tiles against the oracle."""
    n_ind, n_sites = 16000, 64
    idx = [0, 1, 63, 64, 127, 128, n_ind // 2, n_ind - 129, n_ind - 2, n_ind - 1]
    sub = np.concatenate([O.synth_indmajor(7, n_ind, n_sites, miss_frac=0.05, i0=i, n_sub=1) for i in idx])
    with N().Engine(n_ind, n_sites, indep_geno=indep, kernel=kernel, pairwise_del=True) as e:
        e.synth_fill(7, 0.05)
        s, c = e.run()
    so, co = O.all_pairs(sub, pairwise_del=True, indep_geno=indep)
    k = 0
    for a in range(len(idx)):
        for b in range(a + 1, len(idx)):
            g = N().n_pairs(n_ind) - N().n_pairs(n_ind - idx[a]) + (idx[b] - idx[a] - 1)
            assert c[g] == co[k]
            assert abs(s[g] - so[k]) <= 1e-9 * abs(so[k])
            k += 1


def test_random_large_shapes_against_a_two_image_engine():
    """tools/fuzz_large.py, a dozen cases: the sizes where the one-image engine, its fix-up pass, device memory in pieces, staged
    raw uploads and the eager pass are live, each against a two-image engine fed the plain way and the oracle on a few pairs
    (the sweep that found the ring-buffer fault of round 6: profiles/r06_fuzz.txt)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_large.py"), "20001", "12"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "12 cases from seed 20001, 0 bad" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
