"""ngd_score_congruence (host_util.cpp): the score matrix as three weighted squares -- what a single-image engine
(ngd_config.single_image = 2) builds its one operand image from.  Pure host arithmetic: runs without a GPU."""
import numpy as np
import pytest


def N():
    import ngsdist_amd
    return ngsdist_amd


def rebuild(c, d):
    return sum(d[r] * np.outer(c[r], c[r]) for r in range(3))


def dyadic(x, bits=4):
    return np.all(x * (1 << bits) == np.round(x * (1 << bits)))


@pytest.mark.parametrize("avg_nuc_dist", [False, True])
def test_the_references_matrices_give_dyadic_squares(avg_nuc_dist):
    """parse_args.cpp:25-27 and :134-137 (entries 0, 1/2, 1): c and d are small dyadic numbers and give the matrix back
    EXACTLY -- so for called genotypes (one-hot p) every t = c . p, every product d t t' and every sum is exact"""
    S = np.asarray(N().score_matrix(avg_nuc_dist)).reshape(3, 3)
    c, d = N().score_congruence(S)
    assert np.array_equal(rebuild(c, d), S)
    assert dyadic(c) and dyadic(d)
    assert np.count_nonzero(d) == (2 if avg_nuc_dist else 3)  # (--avg_nuc_dist: rank 2)
    # the bilinear form on every pair of called genotypes
    e = np.eye(3)
    for g1 in range(3):
        for g2 in range(3):
            assert sum(d[r] * (c[r] @ e[g1]) * (c[r] @ e[g2]) for r in range(3)) == S[g1, g2]


def test_any_symmetric_matrix_and_no_other():
    rng = np.random.default_rng(1)
    for _ in range(200):
        A = rng.normal(size=(3, 3))
        S = A + A.T
        if rng.integers(0, 3) == 0:
            S[np.diag_indices(3)] = 0  # no square to start from
        if rng.integers(0, 4) == 0:
            S[rng.integers(0, 3)] = 0
            S = np.minimum(S, S.T) * (S != 0) * (S.T != 0)  # a zero row and column: rank below 3
        c, d = N().score_congruence(S)
        assert np.allclose(rebuild(c, d), S, rtol=0, atol=1e-14 * (1 + np.abs(S).max()))
    for S in (np.zeros((3, 3)), np.diag([1.0, 0, -2.0]), np.array([[0, 1.0, 0], [1.0, 0, 0], [0, 0, 0]])):
        c, d = N().score_congruence(S)
        assert np.array_equal(rebuild(c, d), S)
    bad = np.asarray(N().score_matrix(False)).reshape(3, 3).copy()
    bad[0, 1] = 0.25
    with pytest.raises(N().NgdError):
        N().score_congruence(bad)
    with pytest.raises(N().NgdError):
        N().score_congruence(np.full((3, 3), np.nan))
