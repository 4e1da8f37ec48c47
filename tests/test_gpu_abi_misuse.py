"""The C ABI called the way a careless host would: null pointers, geometry that does not fit, calls in the wrong
order, out-of-range options.  Every one of them must come back as a negative code with a message in
ngd_last_error() -- never a crash, an exit or a hang (the host converts codes to error(), the library
itself must not take the process down) -- and the engine must still work afterwards."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_bad_arguments_are_codes_and_the_engine_survives():
    import ngsdist_amd as N
    from ngsdist_amd import _lib
    L = _lib.load()
    vp = C.c_void_p
    dp = C.POINTER(C.c_double)
    u64p = C.POINTER(C.c_uint64)
    u32p = C.POINTER(C.c_uint32)

    def cfg(**kw):
        c = _lib.NgdConfig()
        c.n_ind, c.n_sites = kw.pop("n_ind", 6), kw.pop("n_sites", 40)
        for k, v in enumerate(O.DEFAULT_SCORE):
            c.score[k] = v
        c.indep_geno, c.device, c.shard_world = 1, -1, 1
        for k, v in kw.items():
            setattr(c, k, v)
        return c

    def err(rc):
        assert rc < 0, rc
        assert len(L.ngd_last_error()) > 0

    # ---- ngd_create
    h = vp()
    err(L.ngd_create(None, C.byref(h)))
    err(L.ngd_create(C.byref(cfg()), None))
    for bad in (dict(n_ind=1), dict(n_ind=0), dict(n_sites=0), dict(kernel=99), dict(variant=77), dict(exact_shapes=9),
                dict(shard_rank=3, shard_world=2), dict(n_sites=1 << 40), dict(device=1000), dict(kernel=3, indep_geno=1),
                dict(kernel=1, indep_geno=0), dict(n_ind=1 << 40, n_sites=1 << 40), dict(n_ind=500_000, n_sites=64)):
        h = vp()
        err(L.ngd_create(C.byref(cfg(**bad)), C.byref(h)))
        assert not h.value
    # ---- a good engine, then misuse
    n_ind, n_sites = 6, 40
    h = vp()
    assert L.ngd_create(C.byref(cfg()), C.byref(h)) == 0
    n_pairs = n_ind * (n_ind - 1) // 2
    s = np.zeros(4 * n_pairs)
    c = np.zeros(4 * n_pairs, dtype=np.uint64)
    sp, cp = s.ctypes.data_as(dp), c.ctypes.data_as(u64p)
    err(L.ngd_run(h, None, 0, 0, sp, cp))  # nothing uploaded / committed
    err(L.ngd_run_job_dist(h, None, 0, 0, 0, 0, 1, sp))
    err(L.ngd_commit(None))
    p = O.synth_indmajor(1, n_ind, n_sites)
    err(L.ngd_upload_ind_major(h, None))
    err(L.ngd_upload_sites(h, p.ctypes.data_as(dp), 30, 20))  # 30 + 20 > n_sites
    rc = L.ngd_upload_ind_major(h, p.ctypes.data_as(dp))
    assert rc == 0, (rc, L.ngd_last_error())
    assert L.ngd_commit(h) == 0
    err(L.ngd_upload_ind_major(h, p.ctypes.data_as(dp)))  # write-once
    err(L.ngd_synth_fill(h, 1, 0.0))
    maps = np.array([0, 1, 2, 3], dtype=np.uint64)
    mp = maps.ctypes.data_as(u64p)
    err(L.ngd_run(None, None, 0, 0, sp, cp))
    err(L.ngd_run(h, mp, 4, 0, sp, cp))        # block size 0
    err(L.ngd_run(h, mp, 0, 10, sp, cp))       # no blocks
    err(L.ngd_run(h, mp, 4, 11, sp, cp))       # 44 sites > 40
    err(L.ngd_run(h, mp, 1 << 62, 1 << 62, sp, cp))  # overflowing product
    badmap = np.array([0, 1, 2, 9], dtype=np.uint64)
    err(L.ngd_run(h, badmap.ctypes.data_as(u64p), 4, 10, sp, cp))
    err(L.ngd_run_batch(h, None, 2, 4, 10, sp, cp))
    err(L.ngd_run_batch(h, mp, 0, 4, 10, sp, cp))
    err(L.ngd_run_job(h, None, 2, 4, 10, sp, cp))
    mult = np.ones(4, dtype=np.uint32)
    err(L.ngd_run_mult(h, None, 4, 10, sp, cp))
    err(L.ngd_run_mult(h, mult.ctypes.data_as(u32p), 5, 10, sp, cp))
    err(L.ngd_run_mult_batch(h, None, 1, 4, 10, sp, cp))
    err(L.ngd_run_mult_batch(h, mult.ctypes.data_as(u32p), 0, 4, 10, sp, cp))
    err(L.ngd_run_job_dist(h, mp, 1, 4, 10, 0, 1, None))       # no output
    err(L.ngd_run_job_dist(h, None, 2, 4, 10, 0, 1, sp))        # replicates without block maps
    err(L.ngd_run_job_dist(h, mp, 1, 4, 10, 0, 5, sp))          # evolutionary model 5: "not yet supported"
    err(L.ngd_run_job_dist(h, mp, 1, 4, 11, 0, 1, sp))          # 44 sites > 40
    err(L.ngd_run_job_dist(h, badmap.ctypes.data_as(u64p), 1, 4, 10, 0, 1, sp))
    err(L.ngd_run_job_dist(None, mp, 1, 4, 10, 0, 1, sp))
    err(L.ngd_run_mult_batch_dist(h, None, 1, 4, 10, 0, 1, sp))
    err(L.ngd_run_mult_batch_dist(h, mult.ctypes.data_as(u32p), 0, 4, 10, 0, 1, sp))
    assert L.ngd_run_job_dist(h, mp, 1, 4, 10, 0, 1, sp) == 0   # (and a good one after the bad ones)
    err(L.ngd_run_device(h, None, 0, 0, None, None))
    err(L.ngd_run_job_device(h, mp, 1, 4, 10, None, None))
    err(L.ngd_set_option(h, 9999, 1))
    err(L.ngd_set_option(h, 0, 7))   # NGD_OPT_BOOT_PARTIALS is 0, 1 or 2
    err(L.ngd_set_option(None, 0, 1))
    err(L.ngd_last_timing(h, None))
    err(L.ngd_last_em_work(None, None, None))
    err(L.ngd_drop_caches(None))
    err(L.ngd_finish(None, cp, 3, 0, 1, sp))
    err(L.ngd_finish(sp, cp, 3, 0, 7, sp))
    # ---- still in working order: the plain run and a replicate against the oracle
    assert L.ngd_run(h, None, 0, 0, sp, cp) == 0
    so, co = O.all_pairs(p)
    assert np.array_equal(c[:n_pairs], co) and np.allclose(s[:n_pairs], so, rtol=1e-12)
    assert L.ngd_run(h, mp, 4, 10, sp, cp) == 0
    so, co = O.all_pairs(p, site_src=O.boot_site_src(maps, 10))
    assert np.array_equal(c[:n_pairs], co) and np.allclose(s[:n_pairs], so, rtol=1e-12)
    L.ngd_destroy(h)
    L.ngd_destroy(None)  # a no-op, like free(NULL)


def test_absurd_launch_geometry_is_held_to_the_data_set():
    """ngd_config.n_slices / wg_target far beyond the data set: the engine holds them to what the data allows (at
    least 128 k-groups / one site per slice) instead of allocating a slab per requested slice -- same results"""
    import ngsdist_amd as N
    p = O.synth_indmajor(4, 70, 900)
    for kernel, indep in (("mfma", True), ("em_table", False), ("em_fast", False)):
        so, co = O.all_pairs(p, indep_geno=indep)
        for geom in (dict(n_slices=4_000_000_000), dict(wg_target=4_000_000_000), dict(n_slices=1)):
            with N.Engine(70, 900, indep_geno=indep, kernel=kernel, **geom) as e:
                s, c = e.upload_ind_major(p).commit().run()
                assert e.device_bytes() < 1 << 30
            assert np.array_equal(c, co) and np.max(np.abs(s - so) / np.abs(so)) < 1e-9, (kernel, geom)


def test_engines_give_their_device_memory_back():
    """create / upload / every kind of run (plain, bootstrap by partials and by passes, EM batch) / destroy, twenty times,
    error exits included: the device's free memory comes back to where it was."""
    import ngsdist_amd as N
    from ngsdist_amd import _lib
    L = _lib.load()
    free0, tot = C.c_uint64(), C.c_uint64()

    def free_now():
        assert L.ngd_device_memory(-1, C.byref(free0), C.byref(tot)) == 0
        return free0.value

    n_ind, n_sites, B = 130, 3000, 10
    p = O.synth_indmajor(3, n_ind, n_sites, miss_frac=0.05)
    maps = np.stack([N.Taus(r).block_map(n_sites // B) for r in range(5)])
    with N.Engine(n_ind, n_sites) as e:  # warm the runtime up (its own pools) before taking the baseline
        e.upload_ind_major(p).commit().run()
    base = free_now()
    for it in range(20):
        kernel = ("mfma", "stream", "em_table", "em_fast")[it % 4]
        with N.Engine(n_ind, n_sites, indep_geno=kernel in ("mfma", "stream"), kernel=kernel, pairwise_del=bool(it & 1)) as e:
            e.set_option("boot_partials", it % 3 != 0)
            e.upload_ind_major(p).commit()
            e.run()
            e.run_job(maps, B)
            e.run_job_dist(maps if it % 5 else maps[:1 + it % 3], B, it % 3)  # (pinned buffers, events and copy streams of the engine's)
            e.run(maps[0], B)
        with pytest.raises(N.NgdError):
            N.Engine(n_ind, 1 << 40)  # far too large: the pieces allocated before the failure are freed again
        if it == 0:
            # too large by a factor of two, not of a million: the images are address ranges whose memory arrives later, so
            # ngd_create itself compares what they will take with the device's free memory (NGD_E_NOMEM, -4)
            with pytest.raises(N.NgdError) as big:
                N.Engine(1000, 20_000_000)
            assert big.value.code == -4
        bad = p.copy()
        bad[3, 7, 1] = np.nan
        with N.Engine(n_ind, n_sites) as e:
            e.upload_raw_sites(np.ascontiguousarray(bad.transpose(1, 0, 2)), 0)
            with pytest.raises(N.NgdError):
                e.commit()  # "NaN found!"
    leaked = base - free_now()
    assert leaked < (64 << 20), "device memory not returned: %d MiB" % (leaked >> 20)


def test_nothing_to_fetch_after_a_failed_job():
    """ngd_fetch_matrix serves the matrices of the LAST batch / job call only: a call that fails (here: a block map entry
    out of range, found before anything is launched; then a batch whose buffers cannot be allocated) leaves nothing to
    fetch -- not the matrices of the call before it, whose buffers may have been freed and grown meanwhile"""
    import ngsdist_amd as N
    n_ind, n_sites, B = 40, 600, 4
    p = O.synth_indmajor(3, n_ind, n_sites)
    maps = np.stack([N.Taus(r).block_map(n_sites // B) for r in range(3)])
    with N.Engine(n_ind, n_sites, kernel="mfma") as e:
        e.upload_ind_major(p).commit()
        assert e.run_job_keep(maps, B) == 4
        e.fetch_matrix(3)
        bad = maps.copy()
        bad[1, 5] = n_sites  # >= n_blocks
        with pytest.raises(N.NgdError):
            e.run_job_keep(bad, B)
        for which in (0, 3):
            with pytest.raises(N.NgdError):
                e.fetch_matrix(which)
        e.run_job_dist(maps, B, 1)
        e.fetch_matrix(3)
        with pytest.raises(N.NgdError):
            e.run_job_dist(bad, B, 1)  # (the one-call job: the same rule)
        with pytest.raises(N.NgdError):
            e.fetch_matrix(0)
        assert e.run_job_keep(maps, B) == 4  # and the engine goes on
        s, c = e.fetch_matrix(1)
        s1, c1 = e.run(maps[0], B)
        assert np.array_equal(c, c1) and np.max(np.abs(s - s1) / s1) < 1e-12


@pytest.mark.parametrize("form,shape", [(3, 2 | 1 << 3 | 1 << 6), (5, 1 | 2 << 3), (6, 3 | 3 << 3), (6, 2 | 1 << 3 | 1 << 6)])
def test_a_block_shape_the_kernel_does_not_list_fails_the_run(form, shape, monkeypatch):
    """accum_mfma.hip has one code path per block shape of a form (rows x cols tiles, triangular or not); ngd_create()
    builds only listed shapes (and checks).  Should an unlisted one ever reach the kernel -- forged here through the
    test-only option NGD_OPT_DEBUG_FORGE_JOB -- the block's sums are poisoned and the run fails with NGD_E_HIP, on the
    plain pass and on the per-block partial sums alike; it used to fall through a `default: break` and return zeros."""
    import ngsdist_amd as N
    n_ind, n_sites = 100, 2000
    p = O.synth_indmajor(8, n_ind, n_sites)
    with N.Engine(n_ind, n_sites, kernel="mfma", exact_shapes=form) as e:
        e.upload_ind_major(p).commit()
        s, c = e.run()
        monkeypatch.delenv("NGD_ENABLE_TEST_HOOKS", raising=False)
        with pytest.raises(N.NgdError) as ei:  # a test hook: refused in a process that does not say it is a test
            e.set_option("debug_forge_job", shape)
        assert ei.value.code == -1  # NGD_E_INVALID
        assert np.array_equal(e.run()[0], s)
        monkeypatch.setenv("NGD_ENABLE_TEST_HOOKS", "1")
        e.set_option("debug_forge_job", shape)
        with pytest.raises(N.NgdError) as ei:
            e.run()
        assert ei.value.code == -3  # NGD_E_HIP
        e.set_option("boot_partials", 2)
        with pytest.raises(N.NgdError):
            e.run(N.Taus(1).block_map(n_sites // 8), 8)


def test_device_memory_that_runs_out_behind_ngd_create_is_reported_by_the_first_call_that_needs_it(monkeypatch):
    """Images of a GiB and more are address ranges whose memory a thread of the engine maps 256 MiB at a time behind ngd_create
    (engine.hip dev_alloc_pieces / piece_worker).  A piece that cannot be had -- forced here through the test hook
    NGD_TEST_FAIL_PIECE -- is reported as NGD_E_NOMEM by the first call that needs the memory: the staged upload whose sites
    reach the missing part (the pieces before it serve the sites before it), ngd_upload_*, ngd_commit, ngd_synth_fill; the
    engine is destroyed cleanly and the device memory comes back."""
    import ngsdist_amd as N
    L = N._lib.load()

    def free_now():
        free0, tot0 = C.c_uint64(0), C.c_uint64(0)
        assert L.ngd_device_memory(0, C.byref(free0), C.byref(tot0)) == 0
        return free0.value

    n_ind, n_sites = 600, 120_000  # one image + min(p0, p2): 640 x 120 000 x 24 = 1.8 GB -> 7 pieces, 0.6 GB -> hipMalloc
    with N.Engine(n_ind, n_sites) as e:  # (warm-up: the runtime's own pools)
        e.synth_fill(1).run()
    base = free_now()
    monkeypatch.setenv("NGD_ENABLE_TEST_HOOKS", "1")
    monkeypatch.setenv("NGD_TEST_FAIL_PIECE", "3")
    raw = np.ascontiguousarray(O.synth_indmajor(5, n_ind, 4000).transpose(1, 0, 2))
    for how in ("stage", "upload", "synth", "commit"):
        e = N.Engine(n_ind, n_sites)
        assert e.image_mode() == (2, True)
        with pytest.raises(N.NgdError) as ei:
            if how == "stage":
                e.upload_raw_sites(raw, 0)          # the first 4000 sites lie in the pieces that were mapped ...
                e.upload_raw_sites(raw, 100_000)    # ... these do not
            elif how == "upload":
                e.upload_sites(raw, 0)
            elif how == "synth":
                e.synth_fill(3)
            else:
                e.commit()
        assert ei.value.code == -4 and "piece by piece" in str(ei.value)  # NGD_E_NOMEM
        e.close()
    monkeypatch.delenv("NGD_TEST_FAIL_PIECE")
    with N.Engine(n_ind, n_sites) as e:
        s, c = e.synth_fill(1).run()
        assert np.all(np.isfinite(s))
    leaked = base - free_now()
    assert leaked < (64 << 20), "device memory not returned: %d MiB" % (leaked >> 20)


def test_eager_pass_with_pieces_uploaded_out_of_order_and_small_rings():
    """NGD_OPT_EAGER_FULL only follows a load that arrives in site order; pieces out of order switch it off for the rest of the
    load (what was launched stays valid).  NGD_OPT_STAGE_RING / _PIECE_MIB: the smallest ring, one-MiB pieces."""
    import ngsdist_amd as N
    n_ind, n_sites = 450, 50_000
    p = O.synth_indmajor(9, n_ind, n_sites)
    raw = np.ascontiguousarray(p.transpose(1, 0, 2))
    so, co = O.all_pairs(p, n_threads=16)
    half = n_sites // 2
    for order in ((0, half), (half, 0)):
        with N.Engine(n_ind, n_sites, kernel="mfma") as e:
            e.set_option("stage_piece_mib", 1)
            e.set_option("stage_ring", 2)
            e.set_option("eager_full", 1)
            with pytest.raises(N.NgdError):
                e.set_option("stage_ring", 9)
            for s0 in order:
                e.upload_raw_sites(np.ascontiguousarray(raw[s0:s0 + half]), s0)
            with pytest.raises(N.NgdError):
                e.set_option("eager_full", 0)  # (the ring exists: too late)
            s, c = e.commit().run()
        assert np.array_equal(c, co) and np.max(np.abs(s - so) / np.abs(so)) < 1e-9
