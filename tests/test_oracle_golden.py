"""Pins the CPU oracle (oracle/) before anything is compared with it.

1. against outputs of the reference program kept as data under
   tests/golden/survey_probe/ (provenance: README.md there);
2. against the reference's own em2() compiled from /root/reference into
   oracle/_ref (skipped where that library was never built).
"""
import os

import numpy as np
import pytest

from oracle import oracle as O

SP = os.path.join(os.path.dirname(__file__), "golden", "survey_probe")


def _read(name):
    with open(os.path.join(SP, name)) as fh:
        return fh.read()


@pytest.fixture(scope="module")
def t_gl_raw():
    return np.fromfile(os.path.join(SP, "t_gl.bin"), dtype=np.float64)


def test_indep_model0(t_gl_raw):
    p = O.prep_binary(t_gl_raw, 6, 200)
    assert O.run_reference_flow(p, evol_model=0, indep_geno=True) == _read("t_gl_I0.dist")


def test_em_jc69(t_gl_raw):
    p = O.prep_binary(t_gl_raw, 6, 200)
    assert O.run_reference_flow(p, evol_model=2, indep_geno=False) == _read("t_gl_EM2.dist")


def test_call_geno(t_gl_raw):
    p = O.prep_binary(t_gl_raw, 6, 200, call_geno=True)
    assert O.run_reference_flow(p, evol_model=0, indep_geno=True) == _read("t_gl_CG.dist")
    # called genotypes are exact one-hot vectors -> sums are multiples of 0.5
    s, c = O.all_pairs(p)
    assert np.all(s * 2 == np.round(s * 2))


def test_bootstrap_flow(t_gl_raw):
    p = O.prep_binary(t_gl_raw, 6, 200)
    txt = O.run_reference_flow(p, evol_model=1, indep_geno=True, n_boot_rep=2, boot_block_size=7, seed=12345)
    assert txt == _read("t_gl_B.dist")


def test_text_genotypes():
    p = O.load_text(os.path.join(SP, "t_geno.gz"), 6, 200, in_probs=False)
    assert O.run_reference_flow(p, evol_model=1) == _read("t_T.dist")


def test_identical_individuals_and_pairwise_del():
    p = O.load_text(os.path.join(SP, "id.geno.gz"), 3, 4, in_probs=False)
    assert O.run_reference_flow(p, evol_model=1) == _read("id.dist")
    txt = O.run_reference_flow(p, evol_model=1, pairwise_del=True)
    assert txt == _read("id2.dist")
    assert "-0.0000000000" in txt  # -log(1-0) = -0.0, SURVEY 8a


def test_saturated_pair():
    p = O.load_text(os.path.join(SP, "far.geno.gz"), 2, 2, in_probs=False)
    assert O.run_reference_flow(p, evol_model=1) == _read("far.dist")
    assert "nan" in O.run_reference_flow(p, evol_model=2)


def test_missing_genotype_is_not_exactly_one_third():
    # exp(log(1/3) - logsum) ~ 1/3 but the pair term is 4/9 to ~1e-16
    p = O.load_text(os.path.join(SP, "id.geno.gz"), 3, 4, in_probs=False)
    miss = p[0, 3]
    assert np.allclose(miss, 1 / 3, rtol=1e-15) and O.lib().ngo_miss_data(miss.ctypes.data_as(
        __import__("ctypes").POINTER(__import__("ctypes").c_double))) == 1


@pytest.mark.skipif(O.ref_lib() is None, reason="oracle/_ref not built (reference tree absent)")
def test_em2_bit_identical_to_reference_em2():
    rng = np.random.default_rng(7)
    n = 20000
    a = rng.dirichlet([0.5] * 3, size=n)
    b = rng.dirichlet([0.3] * 3, size=n)
    # edge rows: one-hot, missing, zeros-with-one, tiny values
    a[:4] = [[1, 0, 0], [1 / 3, 1 / 3, 1 / 3], [0, 0, 1], [1e-300, 1 - 1e-12, 1e-12]]
    b[:4] = [[0, 0, 1], [1 / 3, 1 / 3, 1 / 3], [0, 0, 1], [0.5, 0.5, 0]]
    ref = np.empty((n, 9))
    import ctypes as C
    dp = C.POINTER(C.c_double)
    O.ref_lib().ref_em2_batch(n, a.ctypes.data_as(dp), b.ctypes.data_as(dp), ref.ctypes.data_as(dp))
    for k in range(n):
        s, _ = O.em2(a[k], b[k])
        assert np.array_equal(s, ref[k], equal_nan=True), k


@pytest.mark.skipif(O.ref_lib() is None, reason="oracle/_ref not built (reference tree absent)")
def test_pair_loop_on_the_references_own_em2_carries_the_same_bits():
    """oracle.use_reference_em2: the threaded pair loop (ngsDist.cpp:333-364 as restated) calling the reference's own
    em2() -- emOptim2.cpp compiled from where it lies -- instead of the restated one: every sum and count identical,
    missing data, --pairwise_del and a bootstrap replicate included (what bench.py's "reference-em2" baseline times)."""
    p = O.synth_indmajor(21, 9, 700, miss_frac=0.1)
    p[2, 5] = 0.0  # an all-zero site: NaN sums in both
    src = O.boot_site_src(O.Taus(3).block_map(70), 10)
    cases = [dict(), dict(pairwise_del=True), dict(site_src=src, n_sites=700)]
    port = [O.all_pairs(p, indep_geno=False, n_threads=4, **kw) for kw in cases]
    assert O.use_reference_em2(True)
    try:
        ref = [O.all_pairs(p, indep_geno=False, n_threads=4, **kw) for kw in cases]
    finally:
        O.use_reference_em2(False)
    for (s0, c0), (s1, c1) in zip(port, ref):
        assert np.array_equal(s0, s1, equal_nan=True) and np.array_equal(c0, c1)
    assert np.isnan(port[0][0]).any() and not np.isnan(port[0][0]).all()
