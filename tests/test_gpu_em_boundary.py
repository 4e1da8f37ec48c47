"""The EM's stopping rule at its decision boundary (emOptim2.cpp:127, `fabs(lik - oldLik) < tole`).

The rule compares a difference of two logarithms with 0.001, so within a few tens of ulps of an input where that
difference EQUALS the tolerance the decision depends on the last bit of libm's log -- two builds of the reference
itself (another libm, an FMA variant of it) may stop one step apart there.  The device kernels evaluate the same
criterion in other, equally exact forms (ratios of power sums instead of logs), so there they may stop one step
from the oracle too.  What is pinned here:
  * away from a boundary (>= 1e6 ulps on either side) every kernel stops exactly where the oracle does;
  * inside the band every kernel returns one of the two ADJACENT iterates, never anything else;
  * the band is narrow: decisions of the fast forms and of the oracle agree again 4096 ulps from the boundary.
A data set has to hit such a band (relative width ~1e-13) for a site to differ at all; DESIGN.md section 4 prices it.
"""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

SC = O.DEFAULT_SCORE.reshape(3, 3)
KERNELS = ["em_table", "em_fast", "em_faithful"]


def N():
    import ngsdist_amd
    return ngsdist_amd


def norm(v):
    v = np.asarray(v, dtype=np.float64)
    return v / v.sum()


def c_at(g1, g2, T):
    """score-weighted sum of the EM iterate after T steps (closed form of the single-site EM)"""
    f1, f2 = g1 ** T, g2 ** T
    return float((f1 / f1.sum()) @ SC @ (f2 / f2.sum()))


def find_boundary(g1, g2_of, lo, hi):
    """adjacent doubles a < b with different oracle iteration counts"""
    a, b = lo, hi
    na, nb = O.em2(g1, g2_of(a))[1], O.em2(g1, g2_of(b))[1]
    assert na != nb
    while np.nextafter(a, b) < b:
        m = 0.5 * (a + b)
        if O.em2(g1, g2_of(m))[1] == na:
            a = m
        else:
            b, nb = m, O.em2(g1, g2_of(m))[1]
    return a, b, na, nb


def gpu_single_sites(kernel, pairs):
    """one engine, one site, individuals (2k, 2k+1) = k-th probe; returns the k-th pair's sum"""
    n_ind = 2 * len(pairs)
    p = np.zeros((n_ind, 1, 3))
    for k, (g1, g2) in enumerate(pairs):
        p[2 * k, 0], p[2 * k + 1, 0] = g1, g2
    with N().Engine(n_ind, 1, indep_geno=False, kernel=kernel) as e:
        s, c = e.upload_ind_major(p).commit().run()
    idx = [N().n_pairs(n_ind) - N().n_pairs(n_ind - 2 * k) for k in range(len(pairs))]  # pair (2k, 2k+1)
    return s[idx]


CASES = [
    # (g1, second individual as a function of x, bracket): stopping steps around 3, 18 (second table round), 34 (third)
    (norm([0.90, 0.08, 0.02]), lambda x: norm([x, 0.10, 0.05]), (0.5, 3.0)),
    (norm([0.6, 0.3, 0.1]), lambda x: norm([x, 0.25, 0.15]), (0.30, 0.60)),
    (norm([0.3503, 0.3315, 0.3182]), lambda x: norm([x, 0.3305, 0.3124]), (0.352, 0.40)),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_stopping_rule_at_and_around_its_boundary(case):
    g1, g2_of, (lo, hi) = CASES[case]
    a, b, na, nb = find_boundary(g1, g2_of, lo, hi)
    assert abs(na - nb) == 1
    ulp = np.spacing(a)
    near = [a + k * ulp for k in range(-64, 65, 4)]
    far = [a - 4096 * ulp, a + 4096 * ulp, a * (1 - 1e-10), a * (1 + 1e-10), a * (1 - 1e-6), a * (1 + 1e-6)]
    xs = near + far
    pairs = [(g1, g2_of(x)) for x in xs]
    T_or = [O.em2(g1, g2)[1] for g1, g2 in pairs]
    lo_T, hi_T = min(na, nb), max(na, nb)
    for kernel in KERNELS:
        got = gpu_single_sites(kernel, pairs)
        for k, x in enumerate(xs):
            g2 = pairs[k][1]
            cands = {T: c_at(g1, g2, T) for T in (lo_T, hi_T)}
            errs = {T: abs(got[k] - v) / v for T, v in cands.items()}
            T_dev = min(errs, key=errs.get)
            assert errs[T_dev] < 1e-12, (kernel, k, errs)  # one of the two adjacent iterates, nothing else
            if k >= len(near):  # outside the band: the oracle's own stopping step
                assert T_dev == T_or[k], (kernel, x, T_dev, T_or[k])
    # the oracle is itself on one side or the other throughout; the two sides differ by what one more EM step changes
    # (tens of percent of this ONE site's term when the pair's genotypes nearly agree and the term is small) -- the size
    # of a disagreement inside the band, whoever is "right"
    assert set(T_or) <= {lo_T, hi_T}
    assert c_at(g1, g2_of(a), lo_T) != c_at(g1, g2_of(a), hi_T)
