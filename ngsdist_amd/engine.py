"""Host-side handle on the MI355X engine (ctypes over include/ngsdist_amd.h).

The reference is a compiled C++ program, so the product's host is the C++
driver under ngsdist_amd/csrc/host/ (same command line as ngsDist); this module
is the thin Python door the tests and bench.py use.  Names follow the
reference: gen_dist / evol_model / pairwise_del / indep_geno / tot_sites /
boot_block_size (ngsDist.hpp:11-44).
"""
import ctypes as C

import numpy as np

from . import _lib

KERNELS = {"auto": 0, "stream": 1, "mfma": 2, "em_faithful": 3, "em_fast": 4, "em_table": 5}
# NGD_OPT_* of include/ngsdist_amd.h
OPTIONS = {"boot_partials": 1, "boot_max_bytes": 2, "boot_wg": 3, "boot_unaligned": 4, "em_batch": 5,
           "em_spill": 6, "em_spill_bytes": 7, "single_image_bytes": 8, "fixup_work": 9, "stage_piece_mib": 10,
           "stage_ring": 11, "eager_full": 12, "debug_forge_job": 100}

# parse_args.cpp:25-27
DEFAULT_SCORE = (0.0, 0.5, 1.0, 0.5, 0.0, 0.5, 1.0, 0.5, 0.0)


class NgdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ngsdist_amd error %d: %s" % (code, msg))
        self.code = code


def _check(rc):
    if rc != 0:
        raise NgdError(rc, _lib.load().ngd_last_error().decode(errors="replace"))


def device_count():
    return _lib.load().ngd_device_count()


def n_pairs(n_ind):
    return n_ind * (n_ind - 1) // 2


def score_matrix(avg_nuc_dist=False):
    s = list(DEFAULT_SCORE)
    if avg_nuc_dist:  # --avg_nuc_dist, parse_args.cpp:134-137
        s[4] = 0.5
    return s


class Engine:
    """One resident data set on one GPU; `run()` = one replicate's worth of
    gen_dist() over every pair this engine's shard owns."""

    def __init__(self, n_ind, n_sites, score=None, pairwise_del=False, indep_geno=True, kernel="auto",
                 device=-1, shard_rank=0, shard_world=1, variant=0, n_slices=0, wg_target=0, exact_shapes=0,
                 single_image=0, second_image_bytes=0):
        self._L = _lib.load()
        self._h = C.c_void_p()
        cfg = _lib.NgdConfig()
        cfg.n_ind, cfg.n_sites = int(n_ind), int(n_sites)
        sc = DEFAULT_SCORE if score is None else [float(x) for x in np.asarray(score).reshape(9)]
        for k in range(9):
            cfg.score[k] = sc[k]
        cfg.pairwise_del, cfg.indep_geno = int(bool(pairwise_del)), int(bool(indep_geno))
        cfg.device, cfg.kernel = int(device), KERNELS[kernel] if isinstance(kernel, str) else int(kernel)
        cfg.shard_rank, cfg.shard_world = int(shard_rank), int(shard_world)
        # launch geometry, 0 = the engine's defaults (ngd_config)
        cfg.variant, cfg.n_slices, cfg.wg_target, cfg.exact_shapes = int(variant), int(n_slices), int(wg_target), int(exact_shapes)
        # MFMA kernel, operand images: 0 = the engine's choice, 1 / True: one + the other formed per launch, 2: one in
        # congruent coordinates (+ the fix-up pass of nearly identical pairs), 3: two
        cfg.single_image = int(single_image)
        cfg.second_image_mib = int(second_image_bytes) >> 20  # ... except this much of it, kept resident all the same
        self.n_ind, self.n_sites = int(n_ind), int(n_sites)
        self.n_pairs = n_pairs(self.n_ind)
        _check(self._L.ngd_create(C.byref(cfg), C.byref(self._h)))

    # -- lifetime -----------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.ngd_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- input --------------------------------------------------------------
    def upload_ind_major(self, p):
        """p[n_ind][n_sites][3]: in_geno_lkl as gen_dist reads it (normal space)."""
        p = np.ascontiguousarray(p, dtype=np.float64)
        assert p.shape == (self.n_ind, self.n_sites, 3), p.shape
        _check(self._L.ngd_upload_ind_major(self._h, p.ctypes.data_as(C.POINTER(C.c_double))))
        return self

    def upload_sites(self, p, s0=0):
        """p[n][n_ind][3]: a run of sites in the binary file's order."""
        p = np.ascontiguousarray(p, dtype=np.float64)
        assert p.ndim == 3 and p.shape[1:] == (self.n_ind, 3), p.shape
        _check(self._L.ngd_upload_sites(self._h, p.ctypes.data_as(C.POINTER(C.c_double)), int(s0), p.shape[0]))
        return self

    def upload_raw_sites(self, raw, s0=0, in_logscale=False, call_geno=False, N_thresh=0.0, call_thresh=0.0):
        """raw[n][n_ind][3]: doubles as stored in the binary GL file; prepared on the device."""
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        assert raw.ndim == 3 and raw.shape[1:] == (self.n_ind, 3), raw.shape
        pr = _lib.NgdPrep(int(in_logscale), int(call_geno), float(N_thresh), float(call_thresh))
        _check(self._L.ngd_upload_raw_sites(self._h, raw.ctypes.data_as(C.POINTER(C.c_double)), int(s0),
                                            raw.shape[0], C.byref(pr)))
        return self

    def commit(self):
        _check(self._L.ngd_commit(self._h))
        return self

    def synth_fill(self, seed, miss_frac=0.0, site0=0):
        """synthetic data set; site0 = position of this engine's first site in the whole set (site sharding)"""
        _check(self._L.ngd_synth_fill_range(self._h, int(seed), float(miss_frac), int(site0)))
        return self

    # -- the hot path ---------------------------------------------------------
    def _map_args(self, block_map, block_size):
        if block_map is None:
            return None, 0, 0, None
        bm = np.ascontiguousarray(block_map, dtype=np.uint64)
        return bm.ctypes.data_as(C.POINTER(C.c_uint64)), bm.size, int(block_size), bm

    def run(self, block_map=None, block_size=1):
        """-> (sum float64[n_pairs], cnt uint64[n_pairs]) in the reference's pair order."""
        ptr, nb, bs, keep = self._map_args(block_map, block_size)
        s = np.empty(self.n_pairs, dtype=np.float64)
        c = np.empty(self.n_pairs, dtype=np.uint64)
        _check(self._L.ngd_run(self._h, ptr, nb, bs, s.ctypes.data_as(C.POINTER(C.c_double)),
                               c.ctypes.data_as(C.POINTER(C.c_uint64))))
        return s, c

    def run_mult(self, mult, block_size, d_sum_ptr=None, d_cnt_ptr=None):
        """one replicate from the multiplicity of each of this engine's blocks (site sharding)"""
        m = np.ascontiguousarray(mult, dtype=np.uint32)
        mp = m.ctypes.data_as(C.POINTER(C.c_uint32))
        if d_sum_ptr is not None:
            _check(self._L.ngd_run_mult_device(self._h, mp, m.size, int(block_size), C.c_void_p(d_sum_ptr),
                                               C.c_void_p(d_cnt_ptr)))
            return None
        s = np.empty(self.n_pairs, dtype=np.float64)
        c = np.empty(self.n_pairs, dtype=np.uint64)
        _check(self._L.ngd_run_mult(self._h, mp, m.size, int(block_size), s.ctypes.data_as(C.POINTER(C.c_double)),
                                    c.ctypes.data_as(C.POINTER(C.c_uint64))))
        return s, c

    def run_batch(self, block_maps=None, block_size=1, mult=None, d_sum_ptr=None, d_cnt_ptr=None):
        """n_rep bootstrap replicates in one call (ngd_run_batch / ngd_run_mult_batch): block_maps or mult is
        [n_rep][n_blocks]; returns (sum, cnt) of shape [n_rep][n_pairs], or writes them to device buffers."""
        if (block_maps is None) == (mult is None):
            raise ValueError("give block_maps or mult")
        if mult is None:
            a = np.ascontiguousarray(block_maps, dtype=np.uint64)
            ap = a.ctypes.data_as(C.POINTER(C.c_uint64))
            f_host, f_dev = self._L.ngd_run_batch, self._L.ngd_run_batch_device
        else:
            a = np.ascontiguousarray(mult, dtype=np.uint32)
            ap = a.ctypes.data_as(C.POINTER(C.c_uint32))
            f_host, f_dev = self._L.ngd_run_mult_batch, self._L.ngd_run_mult_batch_device
        if a.ndim != 2:
            raise ValueError("expected [n_rep][n_blocks]")
        n_rep, n_blocks = a.shape
        if d_sum_ptr is not None:
            _check(f_dev(self._h, ap, n_rep, n_blocks, int(block_size), C.c_void_p(d_sum_ptr), C.c_void_p(d_cnt_ptr)))
            return None
        s = np.empty((n_rep, self.n_pairs), dtype=np.float64)
        c = np.empty((n_rep, self.n_pairs), dtype=np.uint64)
        _check(f_host(self._h, ap, n_rep, n_blocks, int(block_size), s.ctypes.data_as(C.POINTER(C.c_double)),
                      c.ctypes.data_as(C.POINTER(C.c_uint64))))
        return s, c

    def run_job(self, block_maps=None, block_size=1, d_sum_ptr=None, d_cnt_ptr=None):
        """the whole replicate loop in one call (ngd_run_job): matrix 0 = full data, then one matrix per row of
        block_maps ([n_rep][n_blocks]); returns (sum, cnt) of shape [n_rep + 1][n_pairs]"""
        if block_maps is None or len(block_maps) == 0:
            a, ap, n_rep, n_blocks = None, None, 0, 0
        else:
            a = np.ascontiguousarray(block_maps, dtype=np.uint64)
            if a.ndim != 2:
                raise ValueError("expected [n_rep][n_blocks]")
            ap = a.ctypes.data_as(C.POINTER(C.c_uint64))
            n_rep, n_blocks = a.shape
        if d_sum_ptr is not None:
            _check(self._L.ngd_run_job_device(self._h, ap, n_rep, n_blocks, int(block_size), C.c_void_p(d_sum_ptr),
                                              C.c_void_p(d_cnt_ptr)))
            return None
        s = np.empty((n_rep + 1, self.n_pairs), dtype=np.float64)
        c = np.empty((n_rep + 1, self.n_pairs), dtype=np.uint64)
        _check(self._L.ngd_run_job(self._h, ap, n_rep, n_blocks, int(block_size), s.ctypes.data_as(C.POINTER(C.c_double)),
                                   c.ctypes.data_as(C.POINTER(C.c_uint64))))
        return s, c

    def run_job_dist(self, block_maps=None, block_size=1, evol_model=0, mult=None, out=None, tot_sites=0, lead_full=True):
        """a job and the tail of gen_dist() in one call (ngd_run_job_dist; lead_full=False: ngd_run_batch_dist, mult given:
        ngd_run_mult_batch_dist -- no leading full-data matrix): the finished distances, [n_matrices][n_pairs]; the sums and counts stay in the engine
        (fetch_matrix)"""
        if mult is not None:
            a = np.ascontiguousarray(mult, dtype=np.uint32)
            if a.ndim != 2:
                raise ValueError("expected [n_rep][n_blocks]")
            n_rep, n_blocks = a.shape
            n_mat, fn, ap = n_rep, self._L.ngd_run_mult_batch_dist, a.ctypes.data_as(C.POINTER(C.c_uint32))
        elif block_maps is None or len(block_maps) == 0:
            a, ap, n_rep, n_blocks, n_mat, fn = None, None, 0, 0, 1, self._L.ngd_run_job_dist
        else:
            a = np.ascontiguousarray(block_maps, dtype=np.uint64)
            if a.ndim != 2:
                raise ValueError("expected [n_rep][n_blocks]")
            n_rep, n_blocks = a.shape
            n_mat, fn, ap = n_rep + 1, self._L.ngd_run_job_dist, a.ctypes.data_as(C.POINTER(C.c_uint64))
            if not lead_full:  # replicates only (ngd_run_batch_dist)
                n_mat, fn = n_rep, self._L.ngd_run_batch_dist
        d = np.empty((n_mat, self.n_pairs), dtype=np.float64) if out is None else out
        if d.shape != (n_mat, self.n_pairs) or d.dtype != np.float64 or not d.flags.c_contiguous:
            raise ValueError("out: expected a C-contiguous float64 array [n_matrices][n_pairs]")
        _check(fn(self._h, ap, n_rep, n_blocks, int(block_size), int(tot_sites), int(evol_model), d.ctypes.data_as(C.POINTER(C.c_double))))
        return d

    def run_job_keep(self, block_maps, block_size=1):
        """ngd_run_job with the matrices left in the engine; fetch_matrix(r) copies them out one at a time"""
        a = np.ascontiguousarray(block_maps, dtype=np.uint64)
        n_rep, n_blocks = a.shape
        _check(self._L.ngd_run_job(self._h, a.ctypes.data_as(C.POINTER(C.c_uint64)), n_rep, n_blocks, int(block_size), None, None))
        return n_rep + 1

    def fetch_matrix(self, which):
        s = np.empty(self.n_pairs, dtype=np.float64)
        c = np.empty(self.n_pairs, dtype=np.uint64)
        _check(self._L.ngd_fetch_matrix(self._h, int(which), s.ctypes.data_as(C.POINTER(C.c_double)),
                                        c.ctypes.data_as(C.POINTER(C.c_uint64))))
        return s, c

    def run_device(self, d_sum_ptr, d_cnt_ptr, block_map=None, block_size=1):
        """Results written to caller-owned device buffers (raw addresses)."""
        ptr, nb, bs, keep = self._map_args(block_map, block_size)
        _check(self._L.ngd_run_device(self._h, ptr, nb, bs, C.c_void_p(d_sum_ptr), C.c_void_p(d_cnt_ptr)))

    def set_option(self, name, value):
        """plan selection for the replicate loop (ngd_set_option): boot_partials, boot_max_bytes, boot_wg, boot_unaligned,
        em_batch, em_spill, em_spill_bytes, single_image_bytes, fixup_work"""
        _check(self._L.ngd_set_option(self._h, OPTIONS[name], int(value)))
        return self

    def drop_caches(self):
        """forget the bootstrap block partial sums (benchmarks: charge them to every step)"""
        _check(self._L.ngd_drop_caches(self._h))

    def timing(self):
        t = _lib.NgdTiming()
        _check(self._L.ngd_last_timing(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in t._fields_}

    def spill_timing(self):
        """the spilled-terms plan's accumulation phase kernel by kernel (ngd_last_spill_timing); zeros after another plan"""
        t = _lib.NgdSpillTiming()
        _check(self._L.ngd_last_spill_timing(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in t._fields_}

    def image_mode(self):
        """(what the engine holds: 3 two images / 1 / 2, 0 = not an MFMA engine; whether it recomputes nearly identical pairs)"""
        f = C.c_int(0)
        m = self._L.ngd_image_mode(self._h, C.byref(f))
        return int(m), bool(f.value)

    def fixup(self):
        """the fix-up pass of the last run (ngd_last_fixup): flagged / recomputed / skipped pairs, ms"""
        t = _lib.NgdFixupInfo()
        _check(self._L.ngd_last_fixup(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in t._fields_}

    def em_work(self):
        """table-driven EM kernel: ((tile, site) visits, table rounds) of the last run"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(self._L.ngd_last_em_work(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def shader_clock_mhz(self):
        """shader clock of the last MFMA / table-driven EM launch, sampled inside the kernel (0.0: not sampled)"""
        v = C.c_double(0.0)
        _check(self._L.ngd_last_shader_clock(self._h, C.byref(v)))
        return float(v.value)

    def device_bytes(self):
        return int(self._L.ngd_device_bytes(self._h))


def finish(sum_, cnt, tot_sites=0, evol_model=1, out=None):
    """Tail of gen_dist(), ngsDist.cpp:372-401, on the host's libm.  `out` (float64, same size) avoids a fresh
    allocation per call."""
    L = _lib.load()
    s = np.ascontiguousarray(sum_, dtype=np.float64)
    c = np.ascontiguousarray(cnt, dtype=np.uint64)
    if out is None:
        out = np.empty_like(s)
    elif out.dtype != np.float64 or out.size != s.size or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous float64 array of the same size")
    _check(L.ngd_finish(s.ctypes.data_as(C.POINTER(C.c_double)), c.ctypes.data_as(C.POINTER(C.c_uint64)),
                        s.size, int(tot_sites), int(evol_model), out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def score_congruence(score):
    """the symmetric score matrix as three weighted squares, score = SUM_r d[r] c[r] c[r]^T (ngd_score_congruence: what
    ngd_config.single_image = 2 builds its one operand image from).  Returns (c [3][3], d [3])."""
    L = _lib.load()
    s = np.ascontiguousarray(score, dtype=np.float64).reshape(9)
    c, d = np.zeros(9), np.zeros(3)
    dp = C.POINTER(C.c_double)
    _check(L.ngd_score_congruence(s.ctypes.data_as(dp), c.ctypes.data_as(dp), d.ctypes.data_as(dp)))
    return c.reshape(3, 3), d


def format_matrix(dist, labels, n_threads=0):
    """The print block of one matrix (ngsDist.cpp:282-287) as bytes, from ngd_finish()'s pair-ordered output."""
    L = _lib.load()
    d = np.ascontiguousarray(dist, dtype=np.float64)
    n_ind = len(labels)
    if d.size != n_ind * (n_ind - 1) // 2:
        raise ValueError("dist must hold n_ind*(n_ind-1)/2 cells")
    lab = (C.c_char_p * n_ind)(*[x.encode() if isinstance(x, str) else x for x in labels])
    dp = d.ctypes.data_as(C.POINTER(C.c_double))
    cap = 32 + sum(len(x) + 1 for x in lab) + n_ind * n_ind * 16  # enough unless cells are huge
    while True:
        buf = C.create_string_buffer(cap)
        need = L.ngd_format_matrix(dp, n_ind, lab, buf, cap, int(n_threads))
        if need < 0:
            _check(int(need))
        if need <= cap:
            return buf.raw[:need]
        cap = int(need)


class Taus:
    """gsl_rng_taus as the reference seeds and draws it (ngsDist.cpp:179-180, :421-423)."""

    def __init__(self, seed):
        self._L = _lib.load()
        self._st = (C.c_uint32 * 3)()
        self._L.ngd_taus_seed(self._st, int(seed) & 0xFFFFFFFFFFFFFFFF)

    def get(self):
        return int(self._L.ngd_taus_get(self._st))

    def uniform(self):
        return float(self._L.ngd_taus_uniform(self._st))

    def block_map(self, n_blocks):
        m = np.empty(int(n_blocks), dtype=np.uint64)
        self._L.ngd_boot_block_map(self._st, int(n_blocks), m.ctypes.data_as(C.POINTER(C.c_uint64)))
        return m
