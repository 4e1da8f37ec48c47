"""Parity of the HIP engine (through the C ABI) with the CPU oracle.

Bars (BASELINE.json north_star): bit-exact for called-genotype ("integer")
distances; <= 1e-9 relative for GL / EM floating-point distances.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-9  # north_star tolerance for GL/EM floating point

SP = os.path.join(os.path.dirname(__file__), "golden", "survey_probe")


def N():
    import ngsdist_amd
    return ngsdist_amd


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    den = np.where(b == 0, 1.0, np.abs(b))
    return float(np.max(np.abs(a - b) / den)) if a.size else 0.0


def gpu_pairs(p, kernel, pairwise_del=False, indep_geno=True, score=None, block_map=None, block_size=1,
              ind_major=True):
    n_ind, n_sites, _ = p.shape
    with N().Engine(n_ind, n_sites, score=score, pairwise_del=pairwise_del, indep_geno=indep_geno,
                    kernel=kernel) as e:
        if ind_major:
            e.upload_ind_major(p)
        else:  # site-major in two uneven chunks, like a host streaming the binary file
            sm = np.ascontiguousarray(p.transpose(1, 0, 2))
            cut = max(1, n_sites // 3)
            e.upload_sites(sm[:cut], 0)
            if cut < n_sites:
                e.upload_sites(sm[cut:], cut)
        e.commit()
        return e.run(block_map, block_size)


INDEP_KERNELS = ["stream", "mfma"]
EM_KERNELS = ["em_faithful", "em_fast", "em_table"]


@pytest.mark.parametrize("kernel", INDEP_KERNELS)
@pytest.mark.parametrize("n_ind,n_sites", [(2, 1), (3, 5), (6, 200), (17, 333), (24, 10000), (130, 1000), (257, 515)])
def test_indep_gl(kernel, n_ind, n_sites):
    p = O.synth_indmajor(3, n_ind, n_sites)
    s, c = gpu_pairs(p, kernel)
    so, co = O.all_pairs(p, n_threads=8)
    assert np.array_equal(c, co)
    assert rel_err(s, so) < RTOL


@pytest.mark.parametrize("kernel", INDEP_KERNELS)
def test_indep_pairwise_del_and_upload_by_sites(kernel):
    p = O.synth_indmajor(5, 40, 3000, miss_frac=0.2)
    s, c = gpu_pairs(p, kernel, pairwise_del=True, ind_major=False)
    so, co = O.all_pairs(p, pairwise_del=True, n_threads=8)
    assert np.array_equal(c, co)
    assert c.min() < 3000
    assert rel_err(s, so) < RTOL


@pytest.mark.parametrize("kernel", INDEP_KERNELS)
def test_called_genotypes_bit_exact(kernel):
    """--call_geno / genotype input: every term is a multiple of 0.5 -> any order is exact."""
    rng = np.random.default_rng(1)
    n_ind, n_sites = 24, 10000  # examples/test.sh testA shape
    g = rng.integers(0, 3, size=(n_ind, n_sites))
    p = np.zeros((n_ind, n_sites, 3))
    np.put_along_axis(p, g[..., None], 1.0, axis=2)
    for avg in (False, True):
        sc = O.score_matrix(avg)
        s, c = gpu_pairs(p, kernel, score=sc)
        so, co = O.all_pairs(p, score=sc, n_threads=8)
        assert np.array_equal(s, so) and np.array_equal(c, co)
        for model in (0, 1, 2):
            with np.errstate(all="ignore"):
                d = N().finish(s, c, 0, model)
                do = O.finish(so, co, 0, model)
            assert np.array_equal(d, do, equal_nan=True)  # bit-exact distances


@pytest.mark.parametrize("kernel", INDEP_KERNELS)
def test_golden_t_gl(kernel):
    raw = np.fromfile(os.path.join(SP, "t_gl.bin"), dtype=np.float64)
    p = O.prep_binary(raw, 6, 200)
    s, c = gpu_pairs(p, kernel)
    d = N().finish(s, c, 0, 0)
    txt, _ = O.format_matrix(d, O.default_labels(6))
    assert txt == open(os.path.join(SP, "t_gl_I0.dist")).read()
    pc = O.prep_binary(raw, 6, 200, call_geno=True)
    s, c = gpu_pairs(pc, kernel)
    txt, _ = O.format_matrix(N().finish(s, c, 0, 0), O.default_labels(6))
    assert txt == open(os.path.join(SP, "t_gl_CG.dist")).read()


@pytest.mark.parametrize("kernel", EM_KERNELS)
def test_golden_em(kernel):
    raw = np.fromfile(os.path.join(SP, "t_gl.bin"), dtype=np.float64)
    p = O.prep_binary(raw, 6, 200)
    s, c = gpu_pairs(p, kernel, indep_geno=False)
    txt, _ = O.format_matrix(N().finish(s, c, 0, 2), O.default_labels(6))
    assert txt == open(os.path.join(SP, "t_gl_EM2.dist")).read()


@pytest.mark.parametrize("kernel", EM_KERNELS)
@pytest.mark.parametrize("n_ind,n_sites,miss", [(2, 1, 0.0), (6, 200, 0.0), (33, 700, 0.1), (20, 5000, 0.0),
                                                (65, 300, 0.05), (130, 257, 0.0), (64, 61, 0.0), (128, 33, 0.05),
                                                (193, 24, 0.0)])
def test_em(kernel, n_ind, n_sites, miss):
    p = O.synth_indmajor(9, n_ind, n_sites, miss_frac=miss)
    for pd in (False, True):
        s, c = gpu_pairs(p, kernel, pairwise_del=pd, indep_geno=False)
        so, co = O.all_pairs(p, pairwise_del=pd, indep_geno=False, n_threads=8)
        assert np.array_equal(c, co)
        assert rel_err(s, so) < RTOL


def test_auto_picks_the_em_kernel_by_the_number_of_individuals():
    """kernel = auto on the EM path: the per-pair kernel up to 32 individuals (one mostly empty 64 x 64 tile otherwise),
    the table kernel above -- seen through ngd_last_em_work(), which only the table kernel fills in"""
    for n_ind, table in ((24, False), (32, False), (33, True), (100, True)):
        p = O.synth_indmajor(5, n_ind, 300)
        with N().Engine(n_ind, 300, indep_geno=False, kernel="auto") as e:
            s, c = e.upload_ind_major(p).commit().run()
            assert (e.em_work()[1] > 0) == table, (n_ind, e.em_work())
        so, co = O.all_pairs(p, indep_geno=False, n_threads=8)
        assert np.array_equal(c, co) and rel_err(s, so) < RTOL


@pytest.mark.parametrize("kernel", INDEP_KERNELS + EM_KERNELS)
def test_score_matrix_need_not_be_symmetric(kernel):
    """the ABI takes any 3 x 3 score (params.score, ngsDist.hpp:20): the first individual of a pair indexes its rows
    (ngsDist.cpp:351-353) -- checked with a score that is neither symmetric nor zero on the diagonal"""
    sc = np.array([0.1, 0.7, 1.3, 0.2, 0.05, 0.9, 1.1, 0.4, 0.3])
    p = O.synth_indmajor(77, 70, 300, miss_frac=0.1)
    indep = kernel in INDEP_KERNELS
    for pd in (False, True):
        s, c = gpu_pairs(p, kernel, pairwise_del=pd, indep_geno=indep, score=sc)
        so, co = O.all_pairs(p, score=sc, pairwise_del=pd, indep_geno=indep, n_threads=8)
        assert np.array_equal(c, co) and rel_err(s, so) < RTOL


@pytest.mark.parametrize("kernel", EM_KERNELS)
def test_em_on_degenerate_likelihoods(kernel):
    """EM path on the likelihood vectors real inputs are full of: certain genotypes (1,0,0), exact ties (.5,.5,0),
    missing data (1/3,1/3,1/3), nearly certain ones (1-2e-12, 1e-12, 1e-12) and tiny-but-nonzero entries -- every
    combination of them as a pair, at a site each, against the oracle (their EM stops at step 1 or needs all 50)."""
    pats = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (.5, .5, 0), (0, .5, .5), (.5, 0, .5), (1 / 3, 1 / 3, 1 / 3),
            (1 - 2e-12, 1e-12, 1e-12), (1e-12, 1 - 2e-12, 1e-12), (0.999, 0.001, 1e-300), (0.34, 0.33, 0.33),
            (0.3334, 0.3333, 0.3333), (0.6, 0.4, 1e-200), (0.25, 0.5, 0.25)]
    n = len(pats)
    rng = np.random.default_rng(0)
    n_ind, n_sites = 9, 4 * n * n
    p = np.zeros((n_ind, n_sites, 3))
    for s in range(n_sites):  # individuals 0 and 1 run through every ordered pair of patterns, the others are random picks
        p[0, s], p[1, s] = pats[(s // n) % n], pats[s % n]
        for i in range(2, n_ind):
            p[i, s] = pats[int(rng.integers(0, n))]
    for pd in (False, True):
        s, c = gpu_pairs(p, kernel, pairwise_del=pd, indep_geno=False)
        so, co = O.all_pairs(p, pairwise_del=pd, indep_geno=False, n_threads=8)
        assert np.array_equal(c, co)
        assert np.isfinite(so).all() and rel_err(s, so) < RTOL


@pytest.mark.parametrize("kernel", INDEP_KERNELS + EM_KERNELS)
@pytest.mark.parametrize("block_size", [1, 7, 64])
def test_bootstrap_replicates(kernel, block_size):
    n_ind, n_sites = 12, 1000
    indep = kernel in INDEP_KERNELS
    p = O.synth_indmajor(21, n_ind, n_sites, miss_frac=0.1)
    rng_o, rng_g = O.Taus(12345), N().Taus(12345)
    n_eff = n_sites - n_sites % block_size
    with N().Engine(n_ind, n_sites, pairwise_del=True, indep_geno=indep, kernel=kernel) as e:
        e.upload_ind_major(p).commit()
        for rep in range(3):
            bm_o = rng_o.block_map(n_eff // block_size)
            bm_g = rng_g.block_map(n_eff // block_size)
            assert np.array_equal(bm_o, bm_g)
            s, c = e.run(bm_g, block_size)
            so, co = O.all_pairs(p, pairwise_del=True, indep_geno=indep,
                                 site_src=O.boot_site_src(bm_o, block_size), n_sites=n_eff, n_threads=8)
            assert np.array_equal(c, co)
            assert rel_err(s, so) < RTOL
        # and the full data set again afterwards: bootstrap never moved data
        s, c = e.run()
        so, co = O.all_pairs(p, pairwise_del=True, indep_geno=indep, n_threads=8)
        assert np.array_equal(c, co) and rel_err(s, so) < RTOL


@pytest.mark.parametrize("kernel,block_size", [("mfma", 8), ("mfma", 100), ("mfma", 7), ("mfma", 1), ("mfma", 10),
                                               ("em_fast", 5), ("em_faithful", 12), ("em_table", 5), ("em_table", 32)])
def test_bootstrap_block_partials_equal_weighted_pass(kernel, block_size):
    """Replicates served from per-block partial sums (default) vs. one weighted accumulation
    pass per replicate (option boot_partials = 0): same counts, sums within rounding, both within
    tolerance of the oracle; changing the block size re-derives the partial sums."""
    n_ind, n_sites = 40, 2003
    indep = kernel == "mfma"
    p = O.synth_indmajor(17, n_ind, n_sites, miss_frac=0.15)
    rng = N().Taus(4242)
    with N().Engine(n_ind, n_sites, pairwise_del=True, indep_geno=indep, kernel=kernel) as e:
        e.upload_ind_major(p).commit()
        for B in (block_size, 2 * block_size):
            n_eff = n_sites - n_sites % B
            for rep in range(2):
                bm = rng.block_map(n_eff // B)
                e.set_option("boot_partials", 1)
                s1, c1 = e.run(bm, B)
                t1 = e.timing()
                e.set_option("boot_partials", 0)
                s0, c0 = e.run(bm, B)
                so, co = O.all_pairs(p, pairwise_del=True, indep_geno=indep, site_src=O.boot_site_src(bm, B),
                                     n_sites=n_eff, n_threads=8)
                assert np.array_equal(c1, co) and np.array_equal(c0, co)
                assert rel_err(s1, so) < RTOL and rel_err(s0, so) < RTOL
                assert t1["launches"] == (1 if rep == 0 else 0)  # second replicate re-uses the partial sums
        e.set_option("boot_partials", 1)
        e.drop_caches()
        e.run(bm, B)
        assert e.timing()["launches"] == 1


@pytest.mark.parametrize("kernel,block_size,pdel", [("mfma", 8, True), ("mfma", 100, False), ("mfma", 7, True),
                                                    ("stream", 10, True), ("em_fast", 5, True),
                                                    ("em_faithful", 12, False), ("em_table", 5, True),
                                                    ("em_table", 1, False)])
@pytest.mark.parametrize("n_rep", [1, 3, 21, 37])
def test_bootstrap_batch(kernel, block_size, pdel, n_rep):
    """ngd_run_batch: n_rep replicates in one call == the same replicates one ngd_run at a time (identical
    bits: one summation order for every batch size), each within tolerance of the oracle; covers the
    per-block partial path (mfma with B % 4 == 0, EM) and the per-replicate fallback (mfma B = 7, stream)."""
    n_ind, n_sites = 37, 1501
    indep = kernel in INDEP_KERNELS
    p = O.synth_indmajor(23, n_ind, n_sites, miss_frac=0.2)
    rng = N().Taus(99)
    n_eff = n_sites - n_sites % block_size
    maps = np.stack([rng.block_map(n_eff // block_size) for _ in range(n_rep)])
    with N().Engine(n_ind, n_sites, pairwise_del=pdel, indep_geno=indep, kernel=kernel) as e:
        e.upload_ind_major(p).commit()
        S, Cn = e.run_batch(maps, block_size)
        assert S.shape == (n_rep, e.n_pairs) and Cn.shape == (n_rep, e.n_pairs)
        for r in range(n_rep):
            s1, c1 = e.run(maps[r], block_size)
            assert np.array_equal(Cn[r], c1)
            assert np.array_equal(S[r], s1)
        # the same replicates as multiplicities (site-sharding entry point)
        mult = np.stack([np.bincount(m.astype(np.int64), minlength=maps.shape[1]) for m in maps]).astype(np.uint32)
        S2, C2 = e.run_batch(mult=mult, block_size=block_size)
        assert np.array_equal(S2, S) and np.array_equal(C2, Cn)
    for r in sorted({0, n_rep - 1}):
        so, co = O.all_pairs(p, pairwise_del=pdel, indep_geno=indep, site_src=O.boot_site_src(maps[r], block_size),
                             n_sites=n_eff, n_threads=8)
        assert np.array_equal(Cn[r], co)
        assert rel_err(S[r], so) < RTOL


def test_bootstrap_batch_called_genotypes_bit_exact():
    """called genotypes: every replicate of a batch is exactly the oracle's sum (dyadic terms)."""
    rng = np.random.default_rng(5)
    n_ind, n_sites, B = 30, 4000, 40
    g = rng.integers(0, 3, size=(n_ind, n_sites))
    p = np.zeros((n_ind, n_sites, 3))
    np.put_along_axis(p, g[..., None], 1.0, axis=2)
    t = N().Taus(7)
    maps = np.stack([t.block_map(n_sites // B) for _ in range(10)])
    with N().Engine(n_ind, n_sites, kernel="mfma") as e:
        e.upload_ind_major(p).commit()
        S, Cn = e.run_batch(maps, B)
    for r in (0, 9):
        so, co = O.all_pairs(p, site_src=O.boot_site_src(maps[r], B), n_sites=n_sites, n_threads=8)
        assert np.array_equal(S[r], so) and np.array_equal(Cn[r], co)


@pytest.mark.parametrize("kernel,block_size,pdel,partials,spill", [
    ("mfma", 8, True, True, 1), ("mfma", 7, False, True, 1), ("mfma", 1, True, True, 1), ("stream", 3, False, True, 1),
    ("em_fast", 5, True, True, 1), ("em_fast", 1, True, False, 1), ("em_fast", 7, False, False, 1),
    ("em_faithful", 3, True, False, 1), ("em_faithful", 12, False, True, 1),
    ("em_table", 5, True, True, 1), ("em_table", 1, True, False, 0), ("em_table", 7, False, False, 0),
    ("em_table", 1, True, False, 1), ("em_table", 7, False, False, 1)])
@pytest.mark.parametrize("n_rep", [0, 1, 5, 20])
def test_whole_job_in_one_call(kernel, block_size, pdel, partials, spill, n_rep):
    """ngd_run_job: matrix 0 = ngd_run(NULL) (counts exact, sums to rounding; bit-identical where the plan keeps the
    plain pass), replicates bit-identical to ngd_run(block_map); whatever plan the engine picks: per-block partials
    with the all-ones row (mfma 8 / em with partials on), the EM batch pass (partials off: what large data sets with
    small blocks get; 16 matrices per pass in the per-pair kernels, 8 in the table-driven one), one list-driven
    weighted pass per replicate (mfma 7, 1), the streaming kernel.  The table-driven EM kernel without partials, three
    matrices or more (spill = 1, the default): the terms are spilled once and contracted with every matrix's weights
    (contract_mfma.hip) -- every matrix then agrees with its own pass to rounding, not bit for bit."""
    n_ind, n_sites = 21, 1203
    spilled = kernel == "em_table" and not partials and spill and n_rep + 1 >= 3
    indep = kernel in INDEP_KERNELS
    p = O.synth_indmajor(31, n_ind, n_sites, miss_frac=0.2)
    rng = N().Taus(5)
    n_eff = n_sites - n_sites % block_size
    maps = np.stack([rng.block_map(n_eff // block_size) for _ in range(n_rep)]) if n_rep else None
    with N().Engine(n_ind, n_sites, pairwise_del=pdel, indep_geno=indep, kernel=kernel) as e:
        e.set_option("boot_partials", 1 if partials else 0).set_option("em_spill", spill)
        e.upload_ind_major(p).commit()
        S, Cn = e.run_job(maps, block_size)
        assert S.shape == (n_rep + 1, e.n_pairs)
        s0, c0 = e.run()
        assert np.array_equal(Cn[0], c0) and rel_err(S[0], s0) < 1e-12
        # every plan but the spilled-terms one (the table-driven EM kernel's 8-matrices-per-pass form included) gives
        # a replicate the bits of its own one-replicate pass
        for r in range(n_rep):
            s1, c1 = e.run(maps[r], block_size)
            assert np.array_equal(Cn[r + 1], c1)
            assert rel_err(S[r + 1], s1) < 1e-12 if spilled else np.array_equal(S[r + 1], s1)
        if kernel == "em_table" and not partials and n_rep >= 2 and not spilled:  # the full-data matrix rides along with weight 1
            assert np.array_equal(S[0], s0)
        if n_rep:  # and a batch without the leading matrix
            S2, C2 = e.run_batch(maps, block_size)
            assert np.array_equal(C2, Cn[1:])
            assert rel_err(S2, S[1:]) < 1e-12 if spilled else np.array_equal(S2, S[1:])
    so, co = O.all_pairs(p, pairwise_del=pdel, indep_geno=indep, n_threads=8)
    assert np.array_equal(Cn[0], co) and rel_err(S[0], so) < RTOL
    if n_rep:
        so, co = O.all_pairs(p, pairwise_del=pdel, indep_geno=indep, site_src=O.boot_site_src(maps[-1], block_size),
                             n_sites=n_eff, n_threads=8)
        assert np.array_equal(Cn[-1], co) and rel_err(S[-1], so) < RTOL


@pytest.mark.parametrize("n_ind,n_sites,block_size,n_rep,lead,pdel,scratch_kg", [
    (37, 1501, 1, 5, True, False, 0), (37, 1501, 1, 20, True, True, 7), (70, 803, 3, 40, False, False, 33),
    (21, 1203, 7, 131, True, False, 100), (130, 517, 1, 17, True, True, 64), (9, 40, 10, 3, False, True, 2)])
def test_em_bootstrap_by_spilled_terms_and_one_contraction(n_ind, n_sites, block_size, n_rep, lead, pdel, scratch_kg):
    """EM path, blocks too small for per-block partials (the reference's defaults: parse_args.cpp:29-31): ONE pass of
    the table-driven kernel spills the per-(pair, site) terms chunk by chunk, one FP64 MFMA contraction per chunk adds
    them to every matrix of the job (contract_mfma.hip).  Every matrix <= 1e-12 from its own ngd_run() pass, counts
    exact, first and last matrix against the oracle; chunk sizes down to two k-groups (a partial last k-group, many
    chunks), 3 to 132 matrices (one group of 16, several, more than one batch of 128), sites beyond the last whole
    block (visited by the full-data matrix only)."""
    p = O.synth_indmajor(77, n_ind, n_sites, miss_frac=0.15)
    rng = N().Taus(3)
    n_eff = n_sites - n_sites % block_size
    maps = np.stack([rng.block_map(n_eff // block_size) for _ in range(n_rep)])
    with N().Engine(n_ind, n_sites, pairwise_del=pdel, indep_geno=False, kernel="em_table") as e:
        e.set_option("boot_partials", 0)
        if scratch_kg:  # scratch for that many k-groups (4 sites each) of terms
            n_tiles64 = sum(1 for a in range((n_ind + 63) // 64) for b in range(a, (n_ind + 63) // 64))
            e.set_option("em_spill_bytes", (scratch_kg + 1) * n_tiles64 * 256 * 64 * 8)
        e.upload_ind_major(p).commit()
        S, Cn = e.run_job(maps, block_size) if lead else e.run_batch(maps, block_size)
        e.set_option("em_spill", 0)
        S_b, C_b = e.run_job(maps, block_size) if lead else e.run_batch(maps, block_size)  # 8 matrices per pass
        assert np.array_equal(Cn, C_b) and rel_err(S, S_b) < 1e-12
        for r in sorted({0, 1, n_rep // 2, n_rep - 1}):
            s1, c1 = e.run(maps[r], block_size)
            assert np.array_equal(Cn[r + lead], c1) and rel_err(S[r + lead], s1) < 1e-12
        if lead:
            s0, c0 = e.run()
            assert np.array_equal(Cn[0], c0) and rel_err(S[0], s0) < 1e-12
    for m in sorted({0, n_rep - 1 + lead}):
        src = None if (lead and m == 0) else O.boot_site_src(maps[m - lead], block_size)
        so, co = O.all_pairs(p, pairwise_del=pdel, indep_geno=False, site_src=src,
                             n_sites=n_sites if src is None else n_eff, n_threads=8)
        assert np.array_equal(Cn[m], co) and rel_err(S[m], so) < RTOL


@pytest.mark.parametrize("kernel", ["mfma", "em_table"])
def test_matrices_of_a_job_fetched_one_at_a_time(kernel):
    """ngd_run_job with NULL outputs leaves the matrices in the engine; ngd_fetch_matrix copies them out one by one (what
    the C++ host does: it prints a matrix at a time and holds two n_pairs-long buffers, not n_boot_rep + 1 of them) --
    the same bits as the all-at-once call; an index past the batch, or a fetch after another run, is an error code."""
    n_ind, n_sites, B = 33, 400, 4
    p = O.synth_indmajor(3, n_ind, n_sites, miss_frac=0.1)
    maps = np.stack([N().Taus(r).block_map(n_sites // B) for r in range(5)])
    with N().Engine(n_ind, n_sites, pairwise_del=True, indep_geno=kernel == "mfma", kernel=kernel) as e:
        e.upload_ind_major(p).commit()
        S, Cn = e.run_job(maps, B)
        assert e.run_job_keep(maps, B) == 6
        for r in range(6):
            s, c = e.fetch_matrix(r)
            assert np.array_equal(s, S[r]) and np.array_equal(c, Cn[r])
        with pytest.raises(N().engine.NgdError):
            e.fetch_matrix(6)
        e.run()
        with pytest.raises(N().engine.NgdError):
            e.fetch_matrix(0)


def same_bits(a, b):
    return np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))


@pytest.mark.parametrize("kernel,block_size,pdel,partials,spill", [
    ("mfma", 8, True, True, 1), ("mfma", 8, False, True, 1), ("mfma", 7, False, True, 1), ("mfma", 1, True, False, 1),
    ("stream", 3, False, True, 1), ("em_fast", 5, True, True, 1), ("em_faithful", 3, False, False, 1),
    ("em_table", 5, True, True, 1), ("em_table", 1, False, False, 1), ("em_table", 7, True, False, 0)])
@pytest.mark.parametrize("n_rep", [0, 1, 5, 33, 70])
@pytest.mark.parametrize("evol_model", [0, 2])
def test_job_and_the_tail_of_gen_dist_in_one_call(kernel, block_size, pdel, partials, spill, n_rep, evol_model):
    """ngd_run_job_dist / ngd_run_mult_batch_dist: the job's matrices leave the device in chunks while -- per-block partials --
    the later groups of 32 replicates are still being reduced, and the host finishes each chunk as it lands: the same BITS
    as ngd_run_job followed by ngd_finish on every matrix (ngsDist.cpp:372-401), whatever plan the engine picks, with
    missing data (nan and inf cells among them when a pair has no valid site); the sums and counts stay in the engine."""
    n_ind, n_sites = 23, 903
    indep = kernel in INDEP_KERNELS
    p = O.synth_indmajor(32, n_ind, n_sites, miss_frac=0.25)
    p[3] = p[4] = 1.0 / 3  # missing everywhere (--pairwise_del: pairs without a single valid site -> 0 / 0)
    rng = N().Taus(6)
    n_eff = n_sites - n_sites % block_size
    maps = np.stack([rng.block_map(n_eff // block_size) for _ in range(n_rep)]) if n_rep else None
    with N().Engine(n_ind, n_sites, pairwise_del=pdel, indep_geno=indep, kernel=kernel) as e:
        e.set_option("boot_partials", 1 if partials else 0).set_option("em_spill", spill)
        e.upload_ind_major(p).commit()
        S, Cn = e.run_job(maps, block_size)
        with np.errstate(all="ignore"):
            want = N().finish(S.reshape(-1), Cn.reshape(-1), 0, evol_model).reshape(S.shape)
        for again in range(2):  # (the second call: pinned memory, events and the copy stream are the first call's)
            D = e.run_job_dist(maps, block_size, evol_model)
            assert D.shape == want.shape and same_bits(D, want)
        for r in sorted({0, n_rep}):
            s, c = e.fetch_matrix(r)
            assert same_bits(s, S[r]) and np.array_equal(c, Cn[r])
        if n_rep:  # multiplicities, no leading matrix
            mult = np.stack([np.bincount(m.astype(np.int64), minlength=n_eff // block_size) for m in maps]).astype(np.uint32)
            mult[0] = 0  # a matrix that visits no site: 0 / 0 in every cell
            S2, C2 = e.run_batch(mult=mult, block_size=block_size)
            with np.errstate(all="ignore"):
                want2 = N().finish(S2.reshape(-1), C2.reshape(-1), 0, evol_model).reshape(S2.shape)
            D2 = e.run_job_dist(block_size=block_size, evol_model=evol_model, mult=mult)
            assert same_bits(D2, want2) and np.all(np.isnan(D2[0]))
            S3, C3 = e.run_batch(maps, block_size)  # and block maps without the leading matrix (ngd_run_batch_dist)
            with np.errstate(all="ignore"):
                want3 = N().finish(S3.reshape(-1), C3.reshape(-1), 0, evol_model).reshape(S3.shape)
            assert same_bits(e.run_job_dist(maps, block_size, evol_model, lead_full=False), want3)
        with pytest.raises(N().engine.NgdError):
            e.run_job_dist(maps, block_size, 3)  # (the reference: "model not yet supported")
        if pdel:  # parse_args.cpp:209-210
            with pytest.raises(N().engine.NgdError):
                e.run_job_dist(maps, block_size, evol_model, tot_sites=1000)
        else:  # --tot_sites: the count of every cell (ngsDist.cpp:372-373)
            with np.errstate(all="ignore"):
                want_t = N().finish(S.reshape(-1), Cn.reshape(-1), 1000, evol_model).reshape(S.shape)
            assert same_bits(e.run_job_dist(maps, block_size, evol_model, tot_sites=1000), want_t)
        assert same_bits(e.run_job_dist(maps, block_size, evol_model), want)


@pytest.mark.parametrize("pairwise_del", [False, True])
def test_job_in_one_call_when_the_fixup_pass_patches_the_partial_results(pairwise_del):
    """ngd_run_job_dist on a one-image engine whose data hold nearly identical individuals: the groups' copies are queued
    before the noted pairs are known; once the fix-up pass has patched the per-block partial results the replicates are
    reduced again and EVERY matrix leaves the device again -- nothing of the first copies has been declared landed."""
    n_ind, n_sites, B, n_rep = 400, 4000, 100, 40
    rng = np.random.default_rng(3)
    p = O.synth_indmajor(9, n_ind, n_sites, miss_frac=0.05 if pairwise_del else 0.0)
    g = rng.integers(0, 3, size=n_sites)
    for k in (5, 6, 7, 200):
        q = 1e-12 * (1 + rng.random((n_sites, 3)))
        q[np.arange(n_sites), g] = 0
        q[np.arange(n_sites), g] = 1 - q.sum(axis=1)
        p[k] = q
    maps = np.stack([N().Taus(r + 1).block_map(n_sites // B) for r in range(n_rep)])
    with N().Engine(n_ind, n_sites, pairwise_del=pairwise_del, kernel="mfma", single_image=2) as e:
        e.set_option("boot_partials", 2)
        e.upload_ind_major(p).commit()
        S, Cn = e.run_job(maps, B)
        assert e.fixup()["recomputed"] >= 6
        want = N().finish(S.reshape(-1), Cn.reshape(-1), 0, 1).reshape(S.shape)
        D = e.run_job_dist(maps, B, 1)
        f = e.fixup()
        assert f["recomputed"] >= 6 and f["skipped"] == 0
        assert same_bits(D, want)
    sub = np.array([5, 6, 7, 200, 11])
    so, co = O.all_pairs(np.ascontiguousarray(p[sub]), pairwise_del=pairwise_del, n_threads=8)
    idx = [N()._lib.load().ngd_pair_index(n_ind, int(min(a, b)), int(max(a, b))) for k, a in enumerate(sub) for b in sub[k + 1:]]
    assert np.array_equal(Cn[0][idx], co) and rel_err(S[0][idx], so) < RTOL


@pytest.mark.parametrize("pairwise_del", [False, True])
@pytest.mark.parametrize("block_size,n_sites", [(100, 4000), (8, 1000), (1000, 5000)])
def test_fixup_of_per_block_partial_results_by_one_more_pass(pairwise_del, block_size, n_sites, monkeypatch):
    """a bootstrap job by per-block partial results on a one-image engine whose data hold clusters of nearly identical
    individuals: the noted pairs' slab entries are recomputed tile by tile -- or, where that would cost more, the WHOLE slab once
    more in the two-operand arithmetic from scratch images made a range of whole slices at a time (fixup_partials_by_pass).
    Both routes, forced through the test hook: the same sums as a two-image engine to 1e-9 and the oracle's on the clones."""
    monkeypatch.setenv("NGD_ENABLE_TEST_HOOKS", "1")
    n_ind, n_rep = 400, 5
    rng = np.random.default_rng(block_size)
    p = O.synth_indmajor(9, n_ind, n_sites, miss_frac=0.05 if pairwise_del else 0.0)
    g = rng.integers(0, 3, size=n_sites)
    clones = [5, 6, 7, 200, 201, 399]
    for k in clones:
        q = 1e-12 * (1 + rng.random((n_sites, 3)))
        q[np.arange(n_sites), g] = 0
        q[np.arange(n_sites), g] = 1 - q.sum(axis=1)
        p[k] = q
    n_eff = n_sites - n_sites % block_size
    maps = np.stack([N().Taus(r + 1).block_map(n_eff // block_size) for r in range(n_rep)])
    with N().Engine(n_ind, n_sites, pairwise_del=pairwise_del, kernel="mfma", single_image=3) as e:
        e.set_option("boot_partials", 2)
        S2, C2 = e.upload_ind_major(p).commit().run_job(maps, block_size)
    got = {}
    for route in ("pass", "tiles"):
        monkeypatch.setenv("NGD_TEST_FIX_PARTIALS", route)
        with N().Engine(n_ind, n_sites, pairwise_del=pairwise_del, kernel="mfma", single_image=2) as e:
            e.set_option("boot_partials", 2)
            S, Cn = e.upload_ind_major(p).commit().run_job(maps, block_size)
            f = e.fixup()
        assert f["recomputed"] >= 15 and f["skipped"] == 0 and f["by_pass"] == (1 if route == "pass" else 0), (route, f)
        assert np.array_equal(Cn, C2)
        fin = C2 > 0
        assert rel_err(S[fin], S2[fin]) < RTOL, route
        got[route] = S
    sub = np.array(clones + [11])
    so, co = O.all_pairs(np.ascontiguousarray(p[sub]), pairwise_del=pairwise_del, n_threads=8)
    idx = [N()._lib.load().ngd_pair_index(n_ind, int(min(a, b)), int(max(a, b))) for k, a in enumerate(sub) for b in sub[k + 1:]]
    for route in got:
        assert rel_err(got[route][0][idx], so) < RTOL, route


def test_job_in_one_call_wants_whole_matrices():
    """an engine that owns a share of the pairs holds part of every matrix: the tail cannot be applied to it"""
    p = O.synth_indmajor(1, 140, 64)
    with N().Engine(140, 64, kernel="mfma", shard_rank=0, shard_world=2) as e:
        e.upload_ind_major(p).commit()
        with pytest.raises(N().engine.NgdError):
            e.run_job_dist()


@pytest.mark.parametrize("kernel", ["em_table", "em_fast", "em_faithful"])
def test_em_batch_pass_that_does_not_fit_falls_back_to_one_pass_per_matrix(kernel):
    """the EM batch pass needs RB result planes per slice; when they exceed the scratch budget (NGD_OPT_BOOT_MAX_BYTES,
    or the device) the job is computed one pass per matrix instead -- same bits either way"""
    n_ind, n_sites, B = 40, 600, 3
    p = O.synth_indmajor(8, n_ind, n_sites, miss_frac=0.1)
    maps = np.stack([N().Taus(9 + k).block_map(n_sites // B) for k in range(5)])
    with N().Engine(n_ind, n_sites, indep_geno=False, kernel=kernel, pairwise_del=True) as e:
        e.set_option("boot_partials", 0).set_option("em_spill", 0)
        e.upload_ind_major(p).commit()
        S, Cn = e.run_job(maps, B)
        e.set_option("boot_max_bytes", 4096)
        S2, C2 = e.run_job(maps, B)
    assert np.array_equal(Cn, C2)
    assert np.array_equal(S[1:], S2[1:]) and rel_err(S2[0], S[0]) < 1e-12


def test_heavy_multiplicity_counts():
    """all blocks map to block 0 -> multiplicity n_blocks on a few sites (bit-plane path)."""
    n_ind, n_sites, B = 5, 640, 2
    p = O.synth_indmajor(4, n_ind, n_sites, miss_frac=0.3)
    bm = np.zeros(n_sites // B, dtype=np.uint64)
    s, c = gpu_pairs(p, "mfma", pairwise_del=True, block_map=bm, block_size=B)
    so, co = O.all_pairs(p, pairwise_del=True, site_src=O.boot_site_src(bm, B), n_threads=4)
    assert np.array_equal(c, co) and rel_err(s, so) < RTOL


def test_synth_fill_matches_oracle_generator():
    n_ind, n_sites = 37, 901
    p = O.synth_indmajor(77, n_ind, n_sites, miss_frac=0.05)
    so, co = O.all_pairs(p, pairwise_del=True, n_threads=8)
    for kernel in INDEP_KERNELS:
        with N().Engine(n_ind, n_sites, pairwise_del=True, kernel=kernel) as e:
            s, c = e.synth_fill(77, 0.05).run()
        assert np.array_equal(c, co) and rel_err(s, so) < RTOL


@pytest.mark.parametrize("kernel", ["mfma", "stream", "em_fast", "em_table"])
def test_shards_partition_the_pairs(kernel):
    n_ind, n_sites, world = 300, 256, 3
    p = O.synth_indmajor(8, n_ind, n_sites)
    indep = kernel in INDEP_KERNELS
    tot_s = np.zeros(N().n_pairs(n_ind))
    tot_c = np.zeros(N().n_pairs(n_ind), dtype=np.uint64)
    owned = np.zeros(N().n_pairs(n_ind), dtype=np.int32)
    for r in range(world):
        with N().Engine(n_ind, n_sites, indep_geno=indep, kernel=kernel, shard_rank=r, shard_world=world) as e:
            s, c = e.upload_ind_major(p).commit().run()
        owned += (c > 0)
        tot_s += s
        tot_c += c
    assert np.all(owned == 1)  # disjoint cover -> summing shards == gathering them
    so, co = O.all_pairs(p, indep_geno=indep, n_threads=8)
    assert np.array_equal(tot_c, co) and rel_err(tot_s, so) < RTOL


@pytest.mark.parametrize("n_ind", [17, 33, 64, 130, 200, 256, 383, 600])
@pytest.mark.parametrize("form", [2, 3, 4, 5, 6, 7])
def test_mfma_exact_block_forms(n_ind, form):
    """exact_shapes 2 / 3 / 4 (accum_mfma.hip EXACT): only the MFMA tiles a block needs, in blocks of up to 4 x 4, 2 x 4, or 4 x 4 with a slice's jobs in one workgroup (up to 12 jobs; else form 2):
    tiles of 16 x 16 pairs -- every pair against the oracle; called genotypes bit for bit; a bootstrap replicate as a
    weighted pass, from per-block partials with blocks of 8 sites and with blocks of 6 (masked slices); pair-tile
    shards that partition the pairs (blocks must not straddle a 128-tile).  64 and 256 individuals: the last group of 16
    holds no zero padding (the in-step forms' prefetching wavefront once lost its loop bound to a late-returning
    fragment there: tools/fuzz_parity.py case 40501)."""
    n_sites = 1030
    p = O.synth_indmajor(41 + n_ind, n_ind, n_sites, miss_frac=0.1)
    so, co = O.all_pairs(p, pairwise_del=True, n_threads=8)
    with N().Engine(n_ind, n_sites, pairwise_del=True, kernel="mfma", exact_shapes=form) as e:
        e.upload_ind_major(p).commit()
        s, c = e.run()
        assert np.array_equal(c, co) and rel_err(s, so) < RTOL
        for B, partials in ((8, 1), (6, 1), (5, 0)):
            m = N().Taus(B).block_map(n_sites // B)
            e.set_option("boot_partials", partials)
            s1, c1 = e.run(m, B)
            sb, cb = O.all_pairs(p, pairwise_del=True, site_src=O.boot_site_src(m, B), n_sites=n_sites // B * B, n_threads=8)
            assert np.array_equal(c1, cb) and rel_err(s1, sb) < RTOL, (B, partials)
    rng = np.random.default_rng(n_ind)
    g = rng.integers(0, 3, size=(n_ind, 300))
    pc = np.zeros((n_ind, 300, 3))
    np.put_along_axis(pc, g[..., None], 1.0, axis=2)
    sc, cc = O.all_pairs(pc, n_threads=8)
    tot_s, owned = np.zeros_like(sc), np.zeros(sc.size, dtype=np.int32)
    for r in range(3):
        with N().Engine(n_ind, 300, kernel="mfma", exact_shapes=form, shard_rank=r, shard_world=3) as e:
            s, c = e.upload_ind_major(pc).commit().run()
        owned += (c > 0)
        tot_s += s
    assert np.all(owned == 1) and np.array_equal(tot_s, sc)


@pytest.mark.parametrize("n_ind,form", [(64, 0), (200, 0), (200, 2), (600, 0), (130, 1)])
@pytest.mark.parametrize("scratch_bytes,resident", [(0, 0), (1, 0), (1, 6 << 20)])
def test_mfma_single_image_engine(n_ind, form, scratch_bytes, resident):
    """ngd_config.single_image (engine.hip launch_accumulate): only p is resident, q = score . p is formed for a range of
    k-groups at a time (layout.hip k_qb_range, with emit()'s own arithmetic) while the kernel works the range before.
    Per-block partial sums (blocks of 8 sites and of 6: masked slices) walk ranges of whole slices: the bits of the engine
    that holds both images.  A whole pass gives every slice a piece of every range (a block adds to its plane of the
    slab): other sums than one contiguous run of sites per slice -- equal to rounding, checked against the two-image
    engine and the oracle; called genotypes (exact arithmetic) bit for bit.  scratch_bytes 1: the smallest ranges;
    resident: ngd_config.second_image_mib, the first part of q kept on the device and read where it lies."""
    n_sites = 5000
    p = O.synth_indmajor(77 + n_ind, n_ind, n_sites, miss_frac=0.1)
    rng = np.random.default_rng(n_ind)
    pc = np.zeros((n_ind, n_sites, 3))
    np.put_along_axis(pc, rng.integers(0, 3, size=(n_ind, n_sites))[..., None], 1.0, axis=2)
    out = []
    for single in (3, 1):  # two images / p resident, q formed a range at a time
        with N().Engine(n_ind, n_sites, pairwise_del=True, kernel="mfma", exact_shapes=form, single_image=single,
                        second_image_bytes=resident if single == 1 else 0) as e:
            if single == 1 and scratch_bytes:
                e.set_option("single_image_bytes", scratch_bytes)
            e.upload_ind_major(p).commit()
            r = [e.run()]
            for B, partials in ((8, 2), (6, 2), (7, 0)):  # (2: partials whatever their slab costs -- the same plan in both engines)
                m = N().Taus(B + n_ind).block_map(n_sites // B)
                e.set_option("boot_partials", partials)
                r.append(e.run(m, B))
            r.append(e.run())
            nbytes = e.device_bytes()
        with N().Engine(n_ind, n_sites, kernel="mfma", exact_shapes=form, single_image=single,
                        second_image_bytes=resident if single == 1 else 0) as e:
            if single == 1 and scratch_bytes:
                e.set_option("single_image_bytes", scratch_bytes)
            e.upload_ind_major(pc).commit()
            e.set_option("boot_partials", 0)
            r.append(e.run())
            r.append(e.run(N().Taus(7 + n_ind).block_map(n_sites // 7), 7))
        out.append((r, nbytes))
    (two, bytes_two), (one, bytes_one) = out
    # (bytes_*: a data set this small is two ranges, i.e. no saving -- tests/test_gpu_fullsize.py checks it at cfg 3)
    for k in (1, 2, 5, 6):
        assert np.array_equal(two[k][0], one[k][0]) and np.array_equal(two[k][1], one[k][1]), k
    assert np.array_equal(one[0][0], one[4][0])  # run to run
    for k in (0, 3):
        assert np.array_equal(two[k][1], one[k][1]) and rel_err(one[k][0], two[k][0]) < RTOL
    so, co = O.all_pairs(p, pairwise_del=True, n_threads=8)
    assert np.array_equal(one[0][1], co) and rel_err(one[0][0], so) < RTOL
    m = N().Taus(7 + n_ind).block_map(n_sites // 7)
    sb, cb = O.all_pairs(p, pairwise_del=True, site_src=O.boot_site_src(m, 7), n_sites=n_sites // 7 * 7, n_threads=8)
    assert np.array_equal(one[3][1], cb) and rel_err(one[3][0], sb) < RTOL


@pytest.mark.parametrize("n_ind,form", [(64, 0), (200, 0), (200, 2), (600, 0), (130, 1), (33, 4)])
@pytest.mark.parametrize("avg_nuc_dist", [False, True])
def test_mfma_congruent_single_image_engine(n_ind, form, avg_nuc_dist):
    """ngd_config.single_image = 2 (host_util.cpp ngd_score_congruence()): the score matrix as a sum of three weighted squares,
    score = SUM_r d_r c_r c_r^T, so ONE image t_r = c_r . p serves both operands of the MFMA kernel and d rides on the
    per-index weights (ngsDist.cpp:351-353 regrouped once more).  Both of the reference's matrices
    (parse_args.cpp:25-27, :134-137): every pair against the oracle -- plain pass, a weighted pass, per-block partial
    sums with blocks of 8 sites and of 6 (masked slices), --pairwise_del; called genotypes bit for bit with the two-image
    engine (c, d dyadic: every product and sum exact); half the device memory."""
    n_sites = 3000
    score = O.score_matrix(avg_nuc_dist)
    p = O.synth_indmajor(91 + n_ind, n_ind, n_sites, miss_frac=0.1)
    rng = np.random.default_rng(n_ind)
    pc = np.zeros((n_ind, n_sites, 3))
    np.put_along_axis(pc, rng.integers(0, 3, size=(n_ind, n_sites))[..., None], 1.0, axis=2)
    out = []
    for single in (3, 2):
        with N().Engine(n_ind, n_sites, score=score, pairwise_del=True, kernel="mfma", exact_shapes=form, single_image=single) as e:
            e.upload_ind_major(p).commit()
            r = [e.run()]
            for B, partials in ((8, 2), (6, 2), (7, 0)):
                m = N().Taus(B + n_ind).block_map(n_sites // B)
                e.set_option("boot_partials", partials)
                r.append(e.run(m, B))
            nbytes = e.device_bytes()
        with N().Engine(n_ind, n_sites, score=score, kernel="mfma", exact_shapes=form, single_image=single) as e:
            e.upload_ind_major(pc).commit()
            r.append(e.run())
            for B, partials in ((8, 2), (7, 0)):
                e.set_option("boot_partials", partials)
                r.append(e.run(N().Taus(B + n_ind).block_map(n_sites // B), B))
        out.append((r, nbytes))
    (two, bytes_two), (one, bytes_one) = out
    assert bytes_one < bytes_two
    for k in (4, 5, 6):  # called genotypes
        assert np.array_equal(two[k][0], one[k][0]) and np.array_equal(two[k][1], one[k][1]), k
    so, co = O.all_pairs(p, score=score, pairwise_del=True, n_threads=8)
    assert np.array_equal(one[0][1], co) and rel_err(one[0][0], so) < RTOL
    for k, B in ((1, 8), (2, 6), (3, 7)):
        m = N().Taus(B + n_ind).block_map(n_sites // B)
        sb, cb = O.all_pairs(p, score=score, pairwise_del=True, site_src=O.boot_site_src(m, B), n_sites=n_sites // B * B, n_threads=8)
        assert np.array_equal(one[k][1], cb) and rel_err(one[k][0], sb) < RTOL, k


def clones(n_ind, n_sites, eps, seed=5, n_free=0, hom_only=False):
    """copies of one individual whose likelihoods are confident to `eps` (per-site terms of ~4 eps between two of them);
    the last n_free individuals are ordinary ones.  hom_only: no heterozygous sites (under --avg_nuc_dist two copies of a
    heterozygote are half a difference apart, parse_args.cpp:134-137)"""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 2, size=n_sites) * 2 if hom_only else rng.integers(0, 3, size=n_sites)
    p = eps * (1 + rng.random((n_ind, n_sites, 3)))
    p[:, np.arange(n_sites), g] = 0
    p[:, np.arange(n_sites), g] = 1 - p.sum(axis=2)
    if n_free:
        p[n_ind - n_free:] = O.synth_indmajor(seed, n_free, n_sites)
    return p


@pytest.mark.parametrize("pairwise_del", [False, True])
def test_congruent_single_image_fixup_by_one_more_pass(pairwise_del):
    """So many nearly identical pairs that their tiles would cost more than the whole matrix in the two-image arithmetic (500
    copies x 20 000 sites: 528 tiles): the fix-up goes ONE more pass instead -- P and Q = score . P formed a range of
    k-groups at a time from the image and min(p0, p2) (layout.hip k_pq_range), K1m over the scratch images, the noting rule
    applied once more (reduce.hip k_fix_merge: the noted pairs take the new sums, the others keep their bits).  A plain pass
    and a weighted one (bootstrap multiplicities folded into ONE operand), with 30 ordinary individuals whose pairs must keep
    the one-image bits, against the oracle."""
    n_ind, n_sites, B = 500, 20_000, 16
    p = clones(n_ind, n_sites, 1e-10, n_free=30)
    if pairwise_del:
        miss = np.random.default_rng(4).random((n_ind, n_sites)) < 0.05
        p[miss] = 1.0 / 3
    so, co = O.all_pairs(p, pairwise_del=pairwise_del, n_threads=16)
    m = N().Taus(9).block_map(n_sites // B)
    sb, cb = O.all_pairs(p, pairwise_del=pairwise_del, site_src=O.boot_site_src(m, B), n_threads=16)
    with N().Engine(n_ind, n_sites, pairwise_del=pairwise_del, kernel="mfma") as e:
        assert e.image_mode() == (2, True)
        e.upload_ind_major(p).commit()
        e.set_option("fixup_work", 1)  # a budget of one pair-site: the one-image sums as they are, for the bits of the others
        s0, _ = e.run()
        assert e.fixup()["skipped"] > 0
        e.set_option("fixup_work", 0)
        s, c = e.run()
        f = e.fixup()
        assert f["by_pass"] == 1 and f["skipped"] == 0 and f["recomputed"] == f["flagged"] >= 470 * 469 // 2
        assert np.array_equal(c, co) and rel_err(s, so) < RTOL
        free = np.array([N()._lib.load().ngd_pair_index(n_ind, i, j) for i in range(470, n_ind) for j in range(i + 1, n_ind)])
        assert np.array_equal(s[free], s0[free])  # ordinary pairs: not noted, not touched
        assert rel_err(s0, so) > RTOL  # (the test would prove nothing if the one-image sums were good enough)
        e.set_option("boot_partials", 0)
        s2, c2 = e.run(m, B)
        assert e.fixup()["by_pass"] == 1
        assert np.array_equal(c2, cb) and rel_err(s2, sb) < RTOL


@pytest.mark.parametrize("pairwise_del,n_rep", [(False, 0), (True, 3)])
def test_congruent_single_image_more_noted_pairs_than_the_list_holds(pairwise_del, n_rep):
    """More than 2^20 pairs noted at once (1500 copies of one individual: 1 124 250 pairs): the list of noted pairs
    overflows, WHICH pairs were noted is then unknown, and the pass recomputes every pair of the engine tile by tile -- the
    whole matrix in gen_dist()'s own arithmetic (ngsDist.cpp:351-353).  Rounds 4-5 recomputed none of them.  1e-9 relative
    against the oracle on every pair: a plain pass, and replicates from per-block partial results (many slab slices x
    thousands of tiles: the launches are cut to 2^22 workgroups)."""
    n_ind, n_sites, B = 1500, 48, 8
    p = clones(n_ind, n_sites, 1e-10)
    if pairwise_del:
        miss = np.random.default_rng(3).random((n_ind, n_sites)) < 0.1
        p[miss] = 1.0 / 3
    so, co = O.all_pairs(p, pairwise_del=pairwise_del, n_threads=16)
    with N().Engine(n_ind, n_sites, pairwise_del=pairwise_del, kernel="mfma") as e:  # auto: ONE image above 384 individuals
        assert e.image_mode() == (2, True)
        s, c = e.upload_ind_major(p).commit().run()
        f = e.fixup()
        assert f["flagged"] > 2 ** 20 and f["skipped"] == 0 and f["recomputed"] == N().n_pairs(n_ind)
        assert np.array_equal(c, co) and rel_err(s, so) < RTOL
        if n_rep:
            e.set_option("boot_partials", 2)
            maps = np.stack([N().Taus(40 + r).block_map(n_sites // B) for r in range(n_rep)])
            S, Cn = e.run_batch(maps, B)
            assert e.fixup()["skipped"] == 0
            for r in range(n_rep):
                sb, cb = O.all_pairs(p, pairwise_del=pairwise_del, site_src=O.boot_site_src(maps[r], B), n_threads=16)
                ok = cb > 0  # (a pair without a valid site in the replicate: 0 / 0 on both sides)
                assert np.array_equal(Cn[r], cb) and rel_err(S[r][ok], sb[ok]) < RTOL, r


@pytest.mark.parametrize("avg_nuc_dist", [False, True])
@pytest.mark.parametrize("eps", [1e-9, 1e-13, 1e-22])
def test_congruent_single_image_recomputes_nearly_identical_pairs(avg_nuc_dist, eps):
    """single_image = 2: the three squares carry different signs, so a pair's sum is a difference of terms of order 1 per
    site and its ABSOLUTE error is a few 1e-17 per site -- not enough for copies of one individual with likelihoods
    confident to 1e-9 and less (per-site terms of 4e-9 ... 4e-22), which two images hold to 1e-9 relative.  For the
    reference's matrices the engine therefore recomputes every pair whose sum is below 1e-6 x the sites visited with the
    two-operand arithmetic of ngsDist.cpp:351-353, from p recovered out of the image and min(p0, p2) kept beside it
    (fixup.hip): 1e-9 relative at ANY distance -- a plain pass, a weighted pass, per-block partial sums (aligned and
    masked slices; one and several replicates), --pairwise_del with missing sites -- and only those pairs: the others
    keep the bits of the one-image pass."""
    n_ind, n_sites, n_free = 40, 5000, 8
    score = O.score_matrix(avg_nuc_dist)
    p = clones(n_ind, n_sites, eps, n_free=n_free, hom_only=avg_nuc_dist)
    p[3, 100:140] = 1.0 / 3  # missing sites of one clone (--pairwise_del skips them)
    n_clone_pairs = N().n_pairs(n_ind - n_free)
    so, co = O.all_pairs(p, score=score, pairwise_del=True, n_threads=8)
    assert np.sort(so)[n_clone_pairs - 1] < 1e-4 * (eps / 1e-9) and np.sort(so)[n_clone_pairs] > 100
    with N().Engine(n_ind, n_sites, score=score, pairwise_del=True, kernel="mfma", single_image=2) as e:
        assert e.image_mode() == (2, True)
        e.upload_ind_major(p).commit()
        s1, c1 = e.run()
        f = e.fixup()
        assert f["flagged"] == f["recomputed"] == n_clone_pairs and f["skipped"] == 0
        assert np.array_equal(c1, co) and rel_err(s1, so) < RTOL
        for B, partials, n_rep in ((8, 2, 1), (6, 2, 1), (7, 0, 1), (8, 2, 5), (5, 2, 40)):
            e.set_option("boot_partials", partials)
            maps = np.stack([N().Taus(B + r).block_map(n_sites // B) for r in range(n_rep)])
            S, Cn = e.run_batch(maps, B)
            assert e.fixup()["recomputed"] >= n_clone_pairs
            for r in sorted({0, n_rep - 1}):
                sb, cb = O.all_pairs(p, score=score, pairwise_del=True, site_src=O.boot_site_src(maps[r], B),
                                     n_sites=n_sites // B * B, n_threads=8)
                assert np.array_equal(Cn[r], cb) and rel_err(S[r], sb) < RTOL, (B, partials, n_rep, r)
        S, Cn = e.run_job(np.stack([N().Taus(3).block_map(n_sites // 4)]), 4)  # the whole loop: full data + a replicate
        assert rel_err(S[0], so) < RTOL
    # the same engine without the recomputation would be off by up to 4e-17 per site: a general symmetric matrix gets none
    with N().Engine(n_ind, n_sites, score=score, pairwise_del=True, kernel="mfma", single_image=3) as e:
        s2, _ = e.upload_ind_major(p).commit().run()
        assert e.image_mode() == (3, False) and e.fixup()["flagged"] == 0
    assert rel_err(s2, so) < RTOL
    big = so > 100  # pairs with an ordinary individual: not recomputed
    assert rel_err(s1[big], s2[big]) < 1e-12


@pytest.mark.parametrize("n_sites,n_rep,B,partials", [(1, 0, 1, 0), (3, 4, 1, 2), (5, 2, 5, 1), (40, 3, 4, 2)])
def test_congruent_single_image_fixup_ignores_pairs_without_valid_sites(n_sites, n_rep, B, partials):
    """--pairwise_del with most sites missing (ngsDist.cpp:335-338): thousands of pairs share no valid site and have the sum 0
    -- exactly.  They must not be noted for the fix-up pass: noted, they pass its capacity (4096 pairs) and the pass is
    skipped for the few nearly identical pairs that need it (tools/fuzz_parity.py cases 202651, 202706, 202716, 207606:
    2e-8 relative on the copies).  Under --pairwise_del a pair is noted once its valid-site counts are known: sum below
    1e-6 x ITS OWN count, in any matrix of the job, and never with a count of 0."""
    n_ind = 140
    p = O.synth_indmajor(202716, n_ind, n_sites, miss_frac=0.9)
    p[[3, 17, 60, 61, 99, 120, 139]] = clones(7, n_sites, 1e-9, seed=n_sites)
    so, co = O.all_pairs(p, pairwise_del=True, n_threads=8)
    assert np.count_nonzero(co == 0) > 4096 or n_sites > 5
    with N().Engine(n_ind, n_sites, pairwise_del=True, kernel="mfma", single_image=2) as e:
        e.set_option("boot_partials", partials)
        e.upload_ind_major(p).commit()
        s, c = e.run()
        f = e.fixup()
        assert f["skipped"] == 0 and f["recomputed"] == f["flagged"] >= 21
        assert np.array_equal(c, co) and rel_err(s, so) < RTOL
        if n_rep:
            maps = np.stack([N().Taus(n_sites + r).block_map(n_sites // B) for r in range(n_rep)])
            S, Cn = e.run_batch(maps, B)
            assert e.fixup()["skipped"] == 0
            for r in range(n_rep):
                sb, cb = O.all_pairs(p, pairwise_del=True, site_src=O.boot_site_src(maps[r], B), n_sites=n_sites // B * B, n_threads=8)
                assert np.array_equal(Cn[r], cb) and rel_err(S[r], sb) < RTOL, r


def test_congruent_single_image_leaves_a_data_set_of_clones_alone():
    """The fix-up pass recomputes EVERY noted pair (round 6: no built-in budget; rounds 4-5 left all of them alone above 4.1e9
    pair-sites of work).  Only a caller that SETS a budget (NGD_OPT_FIXUP_WORK) gets the old behaviour: above it nothing is
    recomputed, the sums keep the one-image arithmetic's absolute bound of 4e-17 per site (seven digits below the last one
    %.10f prints), and ngd_last_fixup() says so.  Without one a data set of copies is recomputed whole -- tile by tile
    (clusters: 16 x 16 pairs at a time, fixup.hip k_fixup_tile) and pair by pair (tiles that hold fewer than five noted
    pairs), 1e-9 relative either way.  A symmetric matrix that is not one of the reference's has no fix-up pass at all."""
    n_ind, n_sites = 100, 3000
    p = clones(n_ind, n_sites, 1e-9)
    so, co = O.all_pairs(p, n_threads=8)
    n_tiles = 7 * 8 // 2  # 100 individuals: 7 groups of 16
    with N().Engine(n_ind, n_sites, kernel="mfma", single_image=2) as e:
        e.set_option("fixup_work", int(4.3 * n_tiles * n_sites) - 1)  # one pair-site short of what the 28 tiles cost
        s1, c1 = e.upload_ind_major(p).commit().run()
        f = e.fixup()
        assert f["flagged"] == f["skipped"] == N().n_pairs(n_ind) and f["recomputed"] == 0
        assert np.array_equal(c1, co) and np.max(np.abs(s1 - so)) < 4e-17 * n_sites
        assert np.array_equal(np.round(s1 / n_sites, 10), np.round(so / n_sites, 10))
        e.set_option("fixup_work", 0)  # the default, no budget: all 4950 pairs, 28 tiles
        s1, c1 = e.run()
        f = e.fixup()
    assert f["flagged"] == f["recomputed"] == N().n_pairs(n_ind) and f["skipped"] == 0
    assert np.array_equal(c1, co) and rel_err(s1, so) < RTOL
    # scattered copies: pairs (0, 17), (40, 90), (41, 91) each alone in their tile, a cluster of five in one tile (10 pairs)
    # and the cluster's pairs with individual 70 (another tile, five pairs): tiles and single pairs in one pass, with
    # --pairwise_del, missing sites, a weighted pass and per-block partial results
    n_ind, n_sites = 130, 2000
    p = O.synth_indmajor(5, n_ind, n_sites, miss_frac=0.05)
    for grp in ([0, 17], [40, 90], [41, 91], [100, 101, 102, 103, 104, 70]):
        p[grp] = clones(len(grp), n_sites, 1e-11, seed=grp[0])
    so, co = O.all_pairs(p, pairwise_del=True, n_threads=8)
    with N().Engine(n_ind, n_sites, pairwise_del=True, kernel="mfma", single_image=2) as e:
        s1, c1 = e.upload_ind_major(p).commit().run()
        f = e.fixup()
        assert f["recomputed"] == f["flagged"] >= 3 + 15 and f["skipped"] == 0
        assert np.array_equal(c1, co) and rel_err(s1, so) < RTOL
        for B, partials in ((7, 0), (8, 2)):
            e.set_option("boot_partials", partials)
            m = N().Taus(B).block_map(n_sites // B)
            sb, cb = O.all_pairs(p, pairwise_del=True, site_src=O.boot_site_src(m, B), n_sites=n_sites // B * B, n_threads=8)
            s2, c2 = e.run(m, B)
            assert e.fixup()["recomputed"] >= 18
            assert np.array_equal(c2, cb) and rel_err(s2, sb) < RTOL, (B, partials)
    score = np.array([[0, 0.25, 1], [0.25, 0, 0.5], [1, 0.5, 0.125]])
    p = clones(100, 3000, 1e-9)
    with N().Engine(32, 3000, score=score, kernel="mfma", single_image=2) as e:
        assert e.image_mode() == (2, False)
        s, c = e.upload_ind_major(p[:32]).commit().run()
        assert e.fixup()["flagged"] == 0
    so, co = O.all_pairs(p[:32], score=score, n_threads=8)
    assert np.array_equal(c, co) and rel_err(s, so) < RTOL  # (score[2][2] > 0: copies are not close under this matrix)
    with N().Engine(32, 3000, score=score, kernel="mfma") as e:  # auto: two images for such a matrix
        assert e.image_mode() == (3, False)


def test_auto_holds_one_image_where_memory_matters():
    """ngd_config.single_image = 0: one image in congruent coordinates + the fix-up pass above 384 padded individuals (the
    reference's matrices), two images below (the block forms of a few hundred individuals) and for other matrices"""
    for n_ind, want in ((385, (2, True)), (384, (3, False)), (50, (3, False))):
        with N().Engine(n_ind, 256, kernel="mfma") as e:
            assert e.image_mode() == want, n_ind
    with N().Engine(500, 256, kernel="mfma", score=O.score_matrix(True)) as e:
        assert e.image_mode() == (2, True)
    with N().Engine(500, 256, kernel="mfma", single_image=3) as e:
        assert e.image_mode() == (3, False)
    with N().Engine(500, 256, kernel="stream") as e:
        assert e.image_mode() == (0, False)
    p = clones(500, 2000, 1e-10, n_free=470)
    so, co = O.all_pairs(p, n_threads=8)
    with N().Engine(500, 2000, kernel="mfma") as e:
        s, c = e.upload_ind_major(p).commit().run()
        assert e.fixup()["recomputed"] == N().n_pairs(30)
    assert np.array_equal(c, co) and rel_err(s, so) < RTOL


def test_congruent_single_image_needs_a_symmetric_score_matrix():
    """single_image = 2 rests on a congruence of the score matrix; an asymmetric one is refused (single_image = 1 takes it)"""
    score = O.score_matrix(False).copy()
    score[1] = 0.25  # score[0][1] != score[1][0]
    with pytest.raises(N().NgdError):
        N().Engine(40, 512, score=score, kernel="mfma", single_image=2)
    p = O.synth_indmajor(8, 40, 512)
    so, co = O.all_pairs(p, score=score, n_threads=8)
    with N().Engine(40, 512, score=score, kernel="mfma", single_image=1) as e:
        s, c = e.upload_ind_major(p).commit().run()
    assert np.array_equal(c, co) and rel_err(s, so) < RTOL


def test_single_image_is_an_mfma_engine_option():
    """other kernels hold one image anyway: the flag is accepted and changes nothing; the option that sizes the
    scratch is refused without it, second_image_mib too; with room for the whole second image the engine is the
    two-image one, bit for bit"""
    p = O.synth_indmajor(6, 150, 3000, miss_frac=0.05)
    m = N().Taus(3).block_map(3000 // 5)
    res = []
    for kw in ({}, dict(single_image=True, second_image_bytes=1 << 30)):
        with N().Engine(150, 3000, kernel="mfma", pairwise_del=True, **kw) as e:
            e.upload_ind_major(p).commit()
            res.append((e.run(), e.run(m, 5), e.device_bytes()))
    assert all(np.array_equal(a, b) for k in (0, 1) for a, b in zip(res[0][k], res[1][k])) and res[0][2] == res[1][2]
    with pytest.raises(N().NgdError):
        N().Engine(40, 512, kernel="mfma", second_image_bytes=1 << 20)
    with N().Engine(40, 512, kernel="mfma") as e:
        with pytest.raises(N().NgdError):
            e.set_option("single_image_bytes", 1 << 20)
    p = O.synth_indmajor(5, 40, 512)
    with N().Engine(40, 512, kernel="stream", single_image=True) as e:
        s, c = e.upload_ind_major(p).commit().run()
    with N().Engine(40, 512, kernel="stream") as e:
        s2, c2 = e.upload_ind_major(p).commit().run()
    assert np.array_equal(s, s2) and np.array_equal(c, c2)


@pytest.mark.parametrize("kernel", INDEP_KERNELS + EM_KERNELS)
def test_deterministic_run_to_run(kernel):
    """results must not depend on the run (slabs summed in fixed order, no floating-point atomics) --
    the same engine twice, a second engine, and a bootstrap replicate formed twice: identical bits."""
    n_ind, n_sites = 150, 4096
    p = O.synth_indmajor(2, n_ind, n_sites, miss_frac=0.05)
    indep = kernel in INDEP_KERNELS
    bm = N().Taus(5).block_map(n_sites // 16)
    out = []
    for _ in range(2):
        with N().Engine(n_ind, n_sites, indep_geno=indep, kernel=kernel, pairwise_del=True) as e:
            e.upload_ind_major(p).commit()
            out.append((e.run()[0], e.run()[0], e.run(bm, 16)[0], e.run(bm, 16)[0]))
    a = out[0]
    assert np.array_equal(a[0], a[1]) and np.array_equal(a[2], a[3])
    assert np.array_equal(a[0], out[1][0]) and np.array_equal(a[2], out[1][2])


def test_errors_are_codes_not_exits():
    n = N()
    with pytest.raises(n.NgdError):
        n.Engine(1, 10)
    with pytest.raises(n.NgdError):
        n.Engine(4, 10, indep_geno=True, kernel="em_fast")
    with n.Engine(4, 10) as e:
        with pytest.raises(n.NgdError):
            e.run()  # not committed
        e.upload_ind_major(np.full((4, 10, 3), 1 / 3)).commit()
        with pytest.raises(n.NgdError):
            e.run(np.array([0, 11], dtype=np.uint64), 1)  # map entry out of range
    with pytest.raises(n.NgdError):
        n.finish(np.zeros(1), np.ones(1, dtype=np.uint64), 0, 3)  # K80..TN93: reference error()s


# ---- K0: kernel-input construction on the device (ngd_upload_raw_sites) ----------------------
def _raw_gl(n_ind, n_sites, seed):
    """unnormalised positive likelihood triples, site-major like the binary file, some exact zeros"""
    rng = np.random.default_rng(seed)
    raw = rng.gamma(0.4, size=(n_sites, n_ind, 3)) + 1e-12
    raw[rng.random((n_sites, n_ind)) < 0.05] = [1.0, 0.0, 0.0]          # log(0) -> -inf -> -1e15 clamp
    raw[rng.random((n_sites, n_ind)) < 0.05] = [0.25, 0.25, 0.25]       # missing
    return raw


@pytest.mark.parametrize("kernel,indep", [("mfma", True), ("stream", True), ("em_fast", False), ("em_table", False)])
def test_device_prep_matches_host_prep(kernel, indep):
    n_ind, n_sites = 50, 3001
    raw = _raw_gl(n_ind, n_sites, 5)
    p = O.prep_binary(raw, n_ind, n_sites)
    so, co = O.all_pairs(p, pairwise_del=True, indep_geno=indep, n_threads=8)
    with N().Engine(n_ind, n_sites, pairwise_del=True, indep_geno=indep, kernel=kernel) as e:
        e.upload_raw_sites(raw[:1000], 0).upload_raw_sites(raw[1000:], 1000).commit()
        s, c = e.run()
    assert np.array_equal(c, co)  # missing-site decisions identical
    assert rel_err(s, so) < RTOL


def test_device_prep_log_scale_and_call_geno():
    n_ind, n_sites = 30, 2000
    raw = _raw_gl(n_ind, n_sites, 6)
    with np.errstate(divide="ignore"):
        lraw = np.log(raw)
    lraw[np.isinf(lraw)] = -1e15
    for call in (None, (0.0, 0.0), (0.4, 0.95)):
        kw = dict(call_geno=call is not None, N_thresh=call[0] if call else 0.0, call_thresh=call[1] if call else 0.0)
        p = O.prep_binary(lraw, n_ind, n_sites, in_logscale=True, **kw)
        so, co = O.all_pairs(p, n_threads=8)
        with N().Engine(n_ind, n_sites, kernel="mfma") as e:
            s, c = e.upload_raw_sites(lraw, 0, in_logscale=True, **kw).commit().run()
        assert np.array_equal(c, co)
        assert rel_err(s, so) < RTOL
    # every genotype called (no ties, thresholds 0): one-hot vectors -> dyadic terms -> bit-exact
    rng = np.random.default_rng(8)
    raw2 = rng.gamma(0.4, size=(n_sites, n_ind, 3)) + 1e-9
    p = O.prep_binary(raw2, n_ind, n_sites, call_geno=True)
    so, co = O.all_pairs(p, n_threads=8)
    with N().Engine(n_ind, n_sites, kernel="mfma") as e:
        s, c = e.upload_raw_sites(raw2, 0, call_geno=True).commit().run()
    assert np.array_equal(s, so) and np.array_equal(c, co)


def test_device_prep_reports_nan_like_the_reference():
    raw = _raw_gl(6, 100, 7)
    raw[50, 3, 1] = -0.5  # log of a negative number
    with N().Engine(6, 100) as e:
        e.upload_raw_sites(raw, 0)
        with pytest.raises(N().NgdError) as ei:
            e.commit()
        assert ei.value.code == -6 and "NaN found!" in str(ei.value)
    with N().Engine(6, 100) as e:
        with pytest.raises(N().NgdError):
            e.upload_raw_sites(raw, 0, call_geno=True, N_thresh=0.9, call_thresh=0.5)


# ---- geometry edges: tile boundaries, many tiles, tiny inputs --------------------------------------
@pytest.mark.parametrize("n_ind,n_sites", [(129, 300), (385, 200), (1300, 128), (64, 15), (65, 17), (128, 4)])
def test_tile_boundaries_and_many_tiles(n_ind, n_sites):
    p = O.synth_indmajor(41, n_ind, n_sites, miss_frac=0.1)
    so, co = O.all_pairs(p, pairwise_del=True, n_threads=8)
    s, c = gpu_pairs(p, "mfma", pairwise_del=True)
    assert np.array_equal(c, co) and rel_err(s, so) < RTOL
    if n_ind <= 400:
        so, co = O.all_pairs(p, pairwise_del=True, indep_geno=False, n_threads=8)
        for k in ("em_fast", "em_table"):
            s, c = gpu_pairs(p, k, pairwise_del=True, indep_geno=False)
            assert np.array_equal(c, co) and rel_err(s, so) < RTOL


def test_all_zero_site_vectors_count_and_contribute_nothing():
    """an empty text line leaves (0,0,0): it counts as a site without --pairwise_del and is
    'missing' with it; under EM it makes the pair NaN exactly as on the CPU."""
    p = O.synth_indmajor(3, 9, 64)
    p[2, 10] = 0.0
    p[5, 10] = 0.0
    p[5, 40] = 0.0
    for pd in (False, True):
        s, c = gpu_pairs(p, "mfma", pairwise_del=pd)
        so, co = O.all_pairs(p, pairwise_del=pd)
        assert np.array_equal(c, co) and rel_err(s, so) < RTOL
        for k in EM_KERNELS:
            s, c = gpu_pairs(p, k, pairwise_del=pd, indep_geno=False)
            so, co = O.all_pairs(p, pairwise_del=pd, indep_geno=False)
            assert np.array_equal(c, co)
            assert np.array_equal(np.isnan(s), np.isnan(so))
            ok = ~np.isnan(so)
            assert rel_err(s[ok], so[ok]) < RTOL


@pytest.mark.parametrize("kernel", EM_KERNELS + ["em_table:batch"])
@pytest.mark.parametrize("partials", [0, 1])
def test_all_zero_site_under_em_poisons_only_the_replicates_that_draw_it(kernel, partials):
    """EM path, no --pairwise_del: a (0,0,0) site makes its pairs' sums NaN (0/0 in normalize(), as on the CPU) -- in
    the full-data matrix and in exactly those bootstrap replicates that draw the site's block; a replicate that does
    not draw it must come out finite whichever plan forms it (a weight of zero adds nothing: not 0 x NaN)."""
    n_ind, n_sites, B = 7, 96, 8
    p = O.synth_indmajor(4, n_ind, n_sites)
    p[3, 20] = 0.0  # block 2
    t = N().Taus(11)
    maps = np.stack([t.block_map(n_sites // B) for _ in range(12)])
    assert any(2 in m for m in maps) and any(2 not in m for m in maps)
    # em_table without partials: the spilled-terms plan (its sanitising pass); "em_table:batch": the 8-matrix pass
    kernel, _, how = kernel.partition(":")
    with N().Engine(n_ind, n_sites, indep_geno=False, kernel=kernel) as e:
        e.set_option("boot_partials", partials).set_option("em_spill", 0 if how == "batch" else 1)
        e.upload_ind_major(p).commit()
        S, Cn = e.run_job(maps, B)
        S1 = np.stack([e.run(m, B)[0] for m in maps])
    for r, m in enumerate([None] + list(maps)):
        so, co = O.all_pairs(p, indep_geno=False, site_src=None if m is None else O.boot_site_src(m, B), n_sites=n_sites)
        assert np.array_equal(Cn[r], co)
        assert np.array_equal(np.isnan(S[r]), np.isnan(so)), (r, m)
        ok = ~np.isnan(so)
        assert rel_err(S[r][ok], so[ok]) < RTOL
        if m is not None:
            assert np.array_equal(np.isnan(S1[r - 1]), np.isnan(so))


@pytest.mark.parametrize("spill", [0, 1])
def test_raw_likelihoods_whose_sum_is_denormal_stay_in_the_matrices_that_draw_them(spill):
    """a RAW upload (the reference's prepared input sums to 1) whose three likelihoods sum below 2^-1022: the table
    kernel's reciprocal of that sum is inf and the pair's term not finite -- in the matrices that draw the site, as from
    a one-matrix pass; the matrices that do NOT draw it must stay finite and equal to their own pass (no 0 x NaN), in
    the 8-matrix pass and in the spilled-terms plan alike."""
    n_ind, n_sites, B = 9, 64, 8
    p = O.synth_indmajor(12, n_ind, n_sites)
    p[4, 17] = (3e-310, 1e-310, 2e-310)  # block 2
    t = N().Taus(21)
    maps = np.stack([t.block_map(n_sites // B) for _ in range(10)])
    assert any(2 in m for m in maps) and any(2 not in m for m in maps)
    with N().Engine(n_ind, n_sites, indep_geno=False, kernel="em_table") as e:
        e.set_option("boot_partials", 0).set_option("em_spill", spill)
        e.upload_ind_major(p).commit()
        S, Cn = e.run_job(maps, B)
        for r, m in enumerate(maps):
            s1, c1 = e.run(m, B)
            assert np.array_equal(Cn[r + 1], c1)
            assert np.array_equal(np.isfinite(S[r + 1]), np.isfinite(s1)), (r, m)
            ok = np.isfinite(s1)
            assert (2 in m) == (not ok.all())
            assert rel_err(S[r + 1][ok], s1[ok]) < 1e-12


@pytest.mark.parametrize("kernel", INDEP_KERNELS + EM_KERNELS)
@pytest.mark.parametrize("partials", [0, 1])
def test_replicate_that_draws_none_of_an_engines_blocks(kernel, partials):
    """a site range of a larger job (site sharding, --n_gpus N, --max_device_bytes): some replicates draw NONE of this
    engine's blocks.  Their partial (sum, cnt) must be exactly (0, 0) -- with --pairwise_del the per-pair count of a
    replicate is a weighted popcount, and 'no weight planes' used to mean 'unweighted' (found by tools/fuzz_cli.py)."""
    n_ind, n_sites, B = 9, 40, 10
    p = O.synth_indmajor(31, n_ind, n_sites, miss_frac=0.2)
    mult = np.array([[0, 0, 0, 0], [2, 0, 1, 0], [0, 0, 0, 0], [0, 3, 0, 1], [0, 0, 0, 0]], dtype=np.uint32)
    indep = kernel in INDEP_KERNELS
    for pd in (True, False):
        with N().Engine(n_ind, n_sites, indep_geno=indep, kernel=kernel, pairwise_del=pd) as e:
            e.set_option("boot_partials", partials)
            e.upload_ind_major(p).commit()
            S, Cn = e.run_batch(mult=mult, block_size=B)
            s1 = [e.run_mult(m, B) for m in mult]
        for r, m in enumerate(mult):
            src = np.concatenate([np.repeat(np.arange(b * B, (b + 1) * B), int(k)) for b, k in enumerate(m)] + [np.zeros(0, int)])
            if len(src) == 0:
                assert not S[r].any() and not Cn[r].any(), (kernel, pd, r)
                assert not s1[r][0].any() and not s1[r][1].any(), (kernel, pd, r)
                continue
            so, co = O.all_pairs(p, pairwise_del=pd, indep_geno=indep, site_src=src.astype(np.uint64), n_sites=len(src))
            assert np.array_equal(Cn[r], co) and np.array_equal(s1[r][1], co)
            assert rel_err(S[r], so) < RTOL and rel_err(s1[r][0], so) < RTOL


# ---- site sharding: engines hold contiguous ranges of sites, (sum, cnt) are added -----------------------
@pytest.mark.parametrize("kernel", ["mfma", "em_fast", "em_table", "stream"])
def test_site_shards_add_up(kernel):
    n_ind, n_sites, B, world = 70, 1200, 20, 3
    indep = kernel in INDEP_KERNELS
    p = O.synth_indmajor(23, n_ind, n_sites, miss_frac=0.1)
    cuts = [0, 400, 820, 1200]  # whole blocks of 20 sites per shard
    engines = []
    for r in range(world):
        e = N().Engine(n_ind, cuts[r + 1] - cuts[r], pairwise_del=True, indep_geno=indep, kernel=kernel)
        e.upload_ind_major(p[:, cuts[r]:cuts[r + 1]]).commit()
        engines.append(e)
    try:
        # full data set
        s = sum(e.run()[0] for e in engines)
        c = sum(e.run()[1] for e in engines)
        so, co = O.all_pairs(p, pairwise_del=True, indep_geno=indep, n_threads=8)
        assert np.array_equal(c, co) and rel_err(s, so) < RTOL
        # a bootstrap replicate: the global block map becomes per-shard multiplicities
        rng = N().Taus(77)
        for rep in range(2):
            bm = rng.block_map(n_sites // B)
            mult = np.bincount(bm.astype(np.int64), minlength=n_sites // B).astype(np.uint32)
            tot_s = np.zeros(N().n_pairs(n_ind))
            tot_c = np.zeros(N().n_pairs(n_ind), dtype=np.uint64)
            for r, e in enumerate(engines):
                ms = mult[cuts[r] // B:cuts[r + 1] // B]
                sr, cr = e.run_mult(ms, B)
                tot_s += sr
                tot_c += cr
            so, co = O.all_pairs(p, pairwise_del=True, indep_geno=indep, site_src=O.boot_site_src(bm, B), n_threads=8)
            assert np.array_equal(tot_c, co) and rel_err(tot_s, so) < RTOL
    finally:
        for e in engines:
            e.close()


def test_site_shards_bit_exact_for_called_genotypes_and_synth_ranges():
    n_ind, n_sites = 40, 3000
    rng = np.random.default_rng(3)
    g = rng.integers(0, 3, size=(n_ind, n_sites))
    p = np.zeros((n_ind, n_sites, 3))
    np.put_along_axis(p, g[..., None], 1.0, axis=2)
    so, co = O.all_pairs(p, n_threads=8)
    tot = np.zeros_like(so)
    for lo, hi in ((0, 1111), (1111, 2000), (2000, 3000)):
        with N().Engine(n_ind, hi - lo, kernel="mfma") as e:
            tot += e.upload_ind_major(p[:, lo:hi]).commit().run()[0]
    assert np.array_equal(tot, so)  # dyadic terms: any split of the site axis is exact
    # the synthetic generator addressed by range reproduces the whole set
    ps = O.synth_indmajor(9, n_ind, n_sites)
    so, co = O.all_pairs(ps, n_threads=8)
    tot = np.zeros_like(so)
    for lo, hi in ((0, 1000), (1000, 3000)):
        with N().Engine(n_ind, hi - lo, kernel="mfma") as e:
            tot += e.synth_fill(9, 0.0, site0=lo).run()[0]
    assert rel_err(tot, so) < RTOL


def test_random_shapes_flags_and_plans():
    """a fixed-seed sweep over odd shapes: n_ind / n_sites around the padding edges (16, 64, 128), every kernel,
    random flags, block sizes and replicate counts, partials on/off, EM batch pass on/off, launch geometry left to the
    engine or forced (slice count, exact block shapes) -- ngd_run_job against the oracle."""
    rng = np.random.default_rng(20260)
    kernels = INDEP_KERNELS + EM_KERNELS
    for case in range(35):
        kernel = kernels[case % len(kernels)]
        indep = kernel in INDEP_KERNELS
        n_ind = int(rng.choice([2, 3, 15, 16, 17, 31, 33, 63, 65, 127, 129, 140]))
        n_sites = int(rng.choice([1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 127, 257, 1000, 1025]))
        pdel = bool(rng.integers(0, 2))
        miss = float(rng.choice([0.0, 0.3]))
        B = int(rng.choice([1, 2, 3, 4, 8, 12, 50]))
        B = min(B, n_sites)
        n_rep = int(rng.choice([0, 1, 2, 17]))
        partials, em_batch = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        geom = dict(n_slices=int(rng.choice([0, 0, 1, 3, 8])), exact_shapes=int(rng.integers(0, 3)) if kernel == "mfma" else 0,
                    variant=int(rng.integers(0, 5)) if kernel == "em_table" else 0)
        score = O.score_matrix(bool(rng.integers(0, 2)))
        p = O.synth_indmajor(100 + case, n_ind, n_sites, miss_frac=miss)
        n_eff = n_sites - n_sites % B
        t = N().Taus(case)
        maps = np.stack([t.block_map(n_eff // B) for _ in range(n_rep)]) if n_rep else None
        with N().Engine(n_ind, n_sites, score=score, pairwise_del=pdel, indep_geno=indep, kernel=kernel, **geom) as e:
            e.set_option("boot_partials", partials).set_option("em_batch", em_batch)
            e.upload_ind_major(p).commit()
            S, Cn = e.run_job(maps, B)
        tag = (case, kernel, n_ind, n_sites, pdel, miss, B, n_rep, partials, em_batch, geom)
        for m in sorted({0, n_rep}):
            src = None if m == 0 else O.boot_site_src(maps[m - 1], B)
            so, co = O.all_pairs(p, score=score, pairwise_del=pdel, indep_geno=indep, site_src=src,
                                 n_sites=n_sites if m == 0 else n_eff, n_threads=4)
            assert np.array_equal(Cn[m], co), tag
            assert rel_err(S[m], so) < RTOL, tag


@pytest.mark.parametrize("kernel,indep,n_ind,n_sites", [("mfma", True, 400, 120_000), ("mfma", True, 530, 60_000),
                                                         ("em_table", False, 200, 40_000), ("em_table", False, 90, 30_000)])
@pytest.mark.parametrize("pairwise_del", [False, True])
def test_full_data_pass_started_beside_a_staged_load(kernel, indep, n_ind, n_sites, pairwise_del):
    """NGD_OPT_EAGER_FULL: while the raw chunks of a staged load are still arriving (ngd_stage_acquire / submit: copies and
    the preparation kernel K0 on their streams), the leading slices of the plain full-data pass are accumulated on a
    low-priority stream of their own; the first run() launches only what is left.  Same BITS as the same engine without it
    (a slice's plane of the slab is computed by the same code on the same data whenever it runs), the oracle's values, and a
    second run() -- nothing eager left -- the same again; a bootstrap call first simply drops the eager work."""
    raw = np.ascontiguousarray(O.synth_indmajor(77, n_ind, n_sites, miss_frac=0.05 if pairwise_del else 0.0).transpose(1, 0, 2))
    p = np.ascontiguousarray(raw.transpose(1, 0, 2))
    res = {}
    for eager in (0, 1):
        with N().Engine(n_ind, n_sites, pairwise_del=pairwise_del, indep_geno=indep, kernel=kernel) as e:
            e.set_option("stage_piece_mib", 1)  # many pieces, so that slices complete while later ones are still in flight
            e.set_option("eager_full", eager)
            e.upload_raw_sites(raw, 0).commit()
            s1, c1 = e.run()
            s2, c2 = e.run()
            m = N().Taus(3).block_map(n_sites // 50)
            sb, cb = e.run(m, 50)
            res[eager] = (s1, c1, sb, cb)
            assert np.array_equal(s1, s2) and np.array_equal(c1, c2)
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])
    so, co = O.all_pairs(p, pairwise_del=pairwise_del, indep_geno=indep, n_threads=16)
    assert np.array_equal(res[1][1], co) and rel_err(res[1][0], so) < RTOL
    with N().Engine(n_ind, n_sites, pairwise_del=pairwise_del, indep_geno=indep, kernel=kernel) as e:
        e.set_option("stage_piece_mib", 1)
        e.set_option("eager_full", 1)
        e.upload_raw_sites(raw, 0).commit()
        sb2, cb2 = e.run(m, 50)  # a replicate FIRST: the eager slices are dropped
        assert np.array_equal(sb2, res[0][2]) and np.array_equal(cb2, res[0][3])
        s3, c3 = e.run()
        assert np.array_equal(s3, res[0][0])
