"""CPU checks of the C-ABI library: it loads, exports every symbol the header
declares, and its host-side functions (no GPU involved) agree with the oracle.
No compute entry point is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    os.environ.setdefault("NGD_NO_TORCH", "1")  # the ABI must load without PyTorch
    from ngsdist_amd import _lib
    return _lib.load()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "ngsdist_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ngd_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(lib):
    from ngsdist_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "missing export " + s
    assert sorted(_lib.EXPORTS) == syms, "binding list and header drifted apart"


def test_signatures_are_plain_c():
    txt = open(os.path.join(ROOT, "include", "ngsdist_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)  # declarations only
    for banned in ("torch", "hipStream_t", "std::", "at::Tensor", "#include <hip"):
        assert banned not in txt


def test_abi_version_and_struct_sizes(lib):
    from ngsdist_amd import _lib
    assert lib.ngd_abi_version() == 6  # 6: ngd_run_job_dist / ngd_run_mult_batch_dist (5: ngd_finish_stream, the staging ring's options; 4: ngd_last_spill_timing (3: ngd_config.single_image / second_image_mib in the former reserved words; 2: named launch-geometry fields))
    assert C.sizeof(_lib.NgdConfig) == 8 + 8 + 72 + 4 * 4 + 2 * 4 + 6 * 4
    assert C.sizeof(_lib.NgdTiming) == 4 * 8 + 2 * 8
    assert C.sizeof(_lib.NgdSpillTiming) == 4 * 8 + 9 * 8


def test_no_device_is_an_error_code_not_a_crash(lib):
    import ngsdist_amd as N
    if N.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(N.NgdError) as ei:
        N.Engine(6, 200)
    assert ei.value.code == -2 and "no HIP device" in str(ei.value)


def test_taus_through_the_abi_matches_gsl_known_answer():
    import ngsdist_amd as N
    t = N.Taus(1)
    k = 0
    for _ in range(10000):
        k = t.get()
    assert k == 2733957125  # GSL rng/test.c
    a, b = N.Taus(12345), O.Taus(12345)
    assert [a.uniform() for _ in range(100)] == [b.uniform() for _ in range(100)]
    a, b = N.Taus(7), O.Taus(7)
    for nb in (1, 28, 1000):
        assert np.array_equal(a.block_map(nb), b.block_map(nb))


@pytest.mark.parametrize("model", [0, 1, 2])
def test_finish_matches_oracle_bitwise(model):
    import ngsdist_amd as N
    rng = np.random.default_rng(3)
    cnt = rng.integers(1, 1000, size=500).astype(np.uint64)
    s = rng.random(500) * cnt
    s[:6] = [0.0, cnt[1], 0.75 * cnt[2], 0.9 * cnt[3], 89.5, 0.0]
    cnt[5] = 0  # 0/0
    for tot in (0, 1234):
        with np.errstate(all="ignore"):
            a = N.finish(s, cnt, tot, model)
            b = O.finish(s, cnt, tot, model)
        assert np.array_equal(a, b, equal_nan=True)
        assert np.array_equal(np.signbit(a), np.signbit(b))  # -0.0 and -nan print differently
        assert [O.fmt_cell(x) for x in a[:8]] == [O.fmt_cell(x) for x in b[:8]]


def test_finish_rejects_unimplemented_models():
    import ngsdist_amd as N
    for m in (3, 4, 5, 6, 7):
        with pytest.raises(N.NgdError) as ei:
            N.finish(np.ones(2), np.ones(2, dtype=np.uint64), 0, m)
        assert ei.value.code == -5


def test_pair_index_is_row_major_upper_triangle(lib):
    n = 7
    k = 0
    for i in range(n):
        for j in range(i + 1, n):
            assert lib.ngd_pair_index(n, i, j) == k
            k += 1
    assert lib.ngd_n_pairs(n) == k


def test_shard_of_pair_partitions(lib):
    n = 300  # 3 tile rows -> 6 tiles
    for world in (1, 2, 3, 8):
        owners = {lib.ngd_shard_of_pair(n, i, j, world) for i in (0, 127, 128, 299) for j in range(i + 1, n, 37)}
        assert owners <= set(range(world))
    # one owner per 128 x 128 tile
    assert lib.ngd_shard_of_pair(n, 0, 1, 4) == lib.ngd_shard_of_pair(n, 5, 127, 4)
    assert lib.ngd_shard_of_pair(n, 0, 128, 4) == lib.ngd_shard_of_pair(n, 127, 255, 4)


def test_shard_loads_are_balanced(lib):
    # cfg 3 geometry: 8 tile rows, 28 off-diagonal tiles (cost 64) + 8 diagonal (cost 36)
    n = 1000
    for world in (2, 4, 8):
        load = [0] * world
        for ti in range(8):
            for tj in range(ti, 8):
                r = lib.ngd_shard_of_pair(n, ti * 128, min(n - 1, tj * 128 + 1 if tj > ti else ti * 128 + 1), world)
                load[r] += 36 if ti == tj else 64
        assert max(load) / (sum(load) / world) < 1.03


def test_finish_and_format_from_several_threads_at_once():
    """ngd_finish / ngd_format_matrix share one persistent thread pool; a caller that finds it busy does its work on its
    own thread (the C++ host calls them from one thread per device).  Same bits from every thread."""
    import threading

    import ngsdist_amd as N
    rng = np.random.default_rng(11)
    n = 1 << 18  # above the pool's threshold
    cnt = rng.integers(1, 1000, size=n).astype(np.uint64)
    s = rng.random(n) * cnt * 0.7
    with np.errstate(all="ignore"):
        want = O.finish(s, cnt, 0, 2)
    n_ind = 300
    d = rng.random(n_ind * (n_ind - 1) // 2)
    labels = ["ind%d" % i for i in range(n_ind)]
    want_txt = N.format_matrix(d, labels)
    out, txt = [None] * 6, [None] * 6

    def work(k):
        for _ in range(5):
            with np.errstate(all="ignore"):
                out[k] = N.finish(s, cnt, 0, 2)
            txt[k] = N.format_matrix(d, labels)

    th = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(6):
        assert np.array_equal(out[k], want, equal_nan=True)
        assert txt[k] == want_txt


@pytest.mark.parametrize("model", [0, 1, 2])
def test_finish_stream_works_cells_as_they_land(model, lib):
    """ngd_finish_stream: the tail of gen_dist() (ngsDist.cpp:372-401) over cells that arrive chunk by chunk (a bootstrap job's
    matrices leaving the device): another thread raises the counter of landed cells; cells beyond it hold garbage until then
    and must not be read early.  Same bits as ngd_finish() and as the oracle."""
    import ctypes as C
    import threading
    import time
    rng = np.random.default_rng(model)
    n = 700_001
    s_true = rng.random(n) * 0.3 * 1000
    s_true[:6] = [0.0, 1e-300, 1000.0, 999.9999, 750.0, np.nan]
    c = rng.integers(1, 2000, n).astype(np.uint64)
    c[3] = 0
    with np.errstate(all="ignore"):
        want = O.finish(s_true, c, 0, model)
    s = np.full(n, -7.0)  # what a buffer holds before its chunk lands
    out = np.empty(n)
    landed = C.c_uint64(0)
    dp, up = C.POINTER(C.c_double), C.POINTER(C.c_uint64)

    def feeder():
        edges = [0, 1, 5000, 5001, n // 3, n // 2, n - 1, n]
        for a, b in zip(edges[:-1], edges[1:]):
            time.sleep(0.003)
            s[a:b] = s_true[a:b]
            landed.value = b

    t = threading.Thread(target=feeder)
    t.start()
    rc = lib.ngd_finish_stream(s.ctypes.data_as(dp), c.ctypes.data_as(up), n, 0, model, out.ctypes.data_as(dp), C.byref(landed))
    t.join()
    assert rc == 0
    assert np.array_equal(out.view(np.uint64), want.view(np.uint64))
    assert lib.ngd_finish_stream(s.ctypes.data_as(dp), c.ctypes.data_as(up), n, 0, 3, out.ctypes.data_as(dp), C.byref(landed)) != 0


def test_finish_division_is_vectorised_and_exact():
    """d = sum / cnt is a loop of its own (host_util.cpp divide_range: counts below 2^52 converted by the exponent trick,
    AVX2 clone where the CPU has it): the same bits as the scalar (double)cnt division for every count size and for
    --tot_sites, ranges that start and end off the vector width."""
    import ngsdist_amd as N
    rng = np.random.default_rng(5)
    for n in (1, 3, 4, 5, 63, 4097, 19_900):
        s = rng.random(n) * 1e6 * rng.choice([1e-12, 1.0, 1e9], size=n)
        for cmax in (2, 1 << 20, 1 << 51, (1 << 63) + 12345):
            c = rng.integers(1, cmax, n, dtype=np.uint64)
            if n > 2:
                c[1] = 0
            for tot in (0, 77):
                with np.errstate(all="ignore"):
                    got = N.finish(s, c, tot, 0)
                    want = s / (np.float64(tot) if tot else c.astype(np.float64))
                assert np.array_equal(got.view(np.uint64), np.asarray(want).view(np.uint64)), (n, cmax, tot)
