"""ngd_format_matrix (the print block ngsDist.cpp:282-287): byte-identical to libc's "%.10f" cells.

Runs on the CPU: the formatter is host code of the C-ABI library."""
import ctypes

import numpy as np
import pytest

import ngsdist_amd as N

libc = ctypes.CDLL("libc.so.6")
libc.snprintf.restype = ctypes.c_int
_buf = ctypes.create_string_buffer(512)


def cfmt(x):
    k = libc.snprintf(_buf, 512, b"%.10f", ctypes.c_double(x))
    return _buf.raw[:k]


def expected(vals, labels):
    n = len(labels)
    M = np.zeros((n, n))
    iu = np.triu_indices(n, 1)
    M[iu] = vals
    M.T[iu] = vals  # keeps NaN sign bits, unlike M + M.T
    return b"\n%d\n" % n + b"".join(
        labels[i].encode() + b"".join(b"\t" + cfmt(M[i, j]) for j in range(n)) + b"\n" for i in range(n))


@pytest.mark.parametrize("n,threads", [(2, 0), (7, 1), (65, 3), (130, 0)])
def test_format_matrix_matches_printf(n, threads):
    rng = np.random.default_rng(n)
    npair = n * (n - 1) // 2
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan, 5e-11, 1.5e-10, 2.5e-10, 0.5, 2.0 ** -11,
                        1e300, -1e300, 5e-324, 2.0 ** 29, 2.0 ** 29 - 2.0 ** -24, 0.99999999995, 0.999999999949999,
                        -1e-11, 123456789.123456789])
    pool = np.concatenate([
        rng.random(npair) * 2,                                   # ordinary distances
        10.0 ** rng.uniform(-30, 8, npair),                      # every magnitude
        -rng.random(npair // 2 + 1),                             # negative cells (scores may be negative)
        rng.integers(0, 1 << 40, npair) * 2.0 ** -11 * 1e-3,
        rng.integers(0, 1 << 20, npair) * 2.0 ** -11,            # exact ties at the 11th decimal
        special])
    vals = np.concatenate([special[:min(len(special), npair)], rng.choice(pool, size=max(0, npair - len(special)))])
    vals = vals[:npair]
    labels = ["Ind_%d" % i if i % 3 else "a longer label %d" % i for i in range(n)]
    got = N.format_matrix(vals, labels, n_threads=threads)
    assert got == expected(vals, labels)


@pytest.mark.parametrize("n,threads", [(64, 2), (65, 3), (130, 0), (257, 16), (130, 64)])
@pytest.mark.parametrize("odd_cell", [None, "9.99999999996", "-0.0", "nan", "12.5", "-1e-12"])
def test_format_matrix_of_ordinary_distances_is_formatted_once_per_pair(n, threads, odd_cell):
    """every cell "d.dddddddddd": the fast form (a 12-byte slot per pair, the rows put together from the slots) -- the same
    bytes as printf; ONE other cell anywhere (a value that rounds up to 10.0000000000, a signed zero, nan, 12.5, a negative
    number) sends the whole matrix through the general form"""
    rng = np.random.default_rng(n + threads)
    npair = n * (n - 1) // 2
    vals = np.concatenate([rng.random(npair // 2) * 9.99, 10.0 ** rng.uniform(-15, 0.99, npair - npair // 2 - 4),
                           [0.0, 5e-11, 9.9999999999, 0.99999999995]])
    rng.shuffle(vals)
    if odd_cell is not None:
        vals[int(rng.integers(0, npair))] = float(odd_cell)
    labels = ["Ind_%d" % i if i % 3 else "a longer label %d" % i for i in range(n)]
    for again in range(2):  # (the buffers are kept from call to call)
        assert N.format_matrix(vals, labels, n_threads=threads) == expected(vals, labels)


def test_format_matrix_many_random_cells():
    rng = np.random.default_rng(0)
    n = 400
    vals = rng.random(n * (n - 1) // 2) * rng.choice([1e-6, 1e-3, 1.0, 50.0], size=n * (n - 1) // 2)
    labels = ["I%d" % i for i in range(n)]
    got = N.format_matrix(vals, labels)
    # spot-check 20000 cells against printf rather than all 160000 (keeps the CPU suite short)
    rows = got.split(b"\n")[2:-1]
    assert len(rows) == n
    k = 0
    for i in rng.choice(n, size=50, replace=False):
        cells = rows[i].split(b"\t")
        assert cells[0] == labels[i].encode() and len(cells) == n + 1
        for j in range(n):
            if i == j:
                assert cells[j + 1] == b"0.0000000000"
            else:
                a, b = min(i, j), max(i, j)
                assert cells[j + 1] == cfmt(vals[a * (2 * n - a - 1) // 2 + (b - a - 1)])
                k += 1
    assert k == 50 * (n - 1)


def test_format_matrix_rejects_bad_arguments():
    with pytest.raises(ValueError):
        N.format_matrix(np.zeros(5), ["a", "b", "c"])
