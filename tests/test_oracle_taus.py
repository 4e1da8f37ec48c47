"""gsl_rng_taus restatement: GSL's own published known answer.

GSL rng/test.c: `rng_test (gsl_rng_taus, 1, 10000, 2733957125UL);`
(seed 1, the 10000th output).  GSL is a third-party dependency of the
reference (README.md:20, "gsl v1.15") that is absent from /root/reference and
from this image; the algorithm is restated in oracle/ngsdist_oracle.c and in
the product's host code (ngsdist_amd/csrc/host_util.cpp, checked through the ABI in tests/test_abi.py).
"""
import numpy as np

from oracle import oracle as O


def test_taus_known_answer_seed1():
    t = O.Taus(1)
    k = 0
    for _ in range(10000):
        k = t.get()
    assert k == 2733957125


def test_taus_seed0_is_seed1():
    a, b = O.Taus(0), O.Taus(1)
    assert [a.get() for _ in range(5)] == [b.get() for _ in range(5)]


def test_taus_uniform_seed12345_first5():
    # SURVEY 8c [probe]
    t = O.Taus(12345)
    got = [t.uniform() for _ in range(5)]
    exp = [0.14079645113088191, 0.85450767702423036, 0.54992264253087342,
           0.48398289736360312, 0.38425721903331578]
    assert got == exp


def test_block_map_range_and_stream_continuity():
    t = O.Taus(12345)
    m1 = t.block_map(28)
    m2 = t.block_map(28)
    assert m1.max() < 28 and m2.max() < 28
    u = O.Taus(12345)
    flat = [int(np.floor(u.uniform() * 28)) for _ in range(56)]
    assert list(m1) + list(m2) == flat
