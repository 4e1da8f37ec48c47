"""ctypes door onto oracle/liboracle.so (+ oracle/_ref/libref_em2.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under ngsdist_amd/ imports this.

`run_reference_flow` strings the restated pieces together the way the
reference's main() does (ngsDist.cpp:156-289): load -> prep -> replicate loop
(bootstrap map, all pairs, finish, print).
"""
import ctypes as C
import gzip
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

DEFAULT_SCORE = np.array([0, .5, 1, .5, 0, .5, 1, .5, 0], dtype=np.float64)  # parse_args.cpp:25-27


def score_matrix(avg_nuc_dist=False):
    s = DEFAULT_SCORE.copy()
    if avg_nuc_dist:  # parse_args.cpp:134-137
        s[4] = 0.5
    return s


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        u64p = C.POINTER(C.c_uint64)
        L.ngo_taus_set.argtypes = [C.c_void_p, C.c_uint64]
        L.ngo_taus_get.argtypes = [C.c_void_p]
        L.ngo_taus_get.restype = C.c_uint32
        L.ngo_taus_uniform.argtypes = [C.c_void_p]
        L.ngo_taus_uniform.restype = C.c_double
        L.ngo_boot_block_map.argtypes = [C.c_void_p, C.c_uint64, u64p]
        L.ngo_prep_binary.argtypes = [dp, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, dp]
        L.ngo_prep_binary.restype = C.c_int
        L.ngo_prep_text_probs_one.argtypes = [dp, C.c_int, C.c_int, C.c_double, C.c_double, dp]
        L.ngo_prep_text_geno_one.argtypes = [C.c_double, dp]
        L.ngo_prep_text_geno_one.restype = C.c_int
        L.ngo_prep_empty_line_one.argtypes = [C.c_int, C.c_double, C.c_double, dp]
        L.ngo_miss_data.argtypes = [dp]
        L.ngo_miss_data.restype = C.c_int
        L.ngo_em2.argtypes = [dp, dp, dp, C.c_double, C.c_int]
        L.ngo_em2.restype = C.c_int
        L.ngo_pair_accum.argtypes = [dp, C.c_uint64, u64p, C.c_uint64, C.c_uint64, C.c_uint64, dp,
                                     C.c_int, C.c_int, dp, u64p, u64p]
        L.ngo_all_pairs.argtypes = [dp, C.c_uint64, C.c_uint64, u64p, C.c_uint64, dp, C.c_int, C.c_int,
                                    C.c_int, dp, u64p, u64p]
        L.ngo_all_pairs.restype = C.c_int
        L.ngo_finish.argtypes = [C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_int)]
        L.ngo_finish.restype = C.c_double
        L.ngo_format_cell.argtypes = [C.c_double, C.c_char_p, C.c_int]
        L.ngo_format_cell.restype = C.c_int
        L.ngo_synth_one.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, dp]
        L.ngo_synth_fill_indmajor.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                              C.c_uint64, C.c_double, dp]
        _LIB = L
    return _LIB


def ref_lib():
    """The reference's own em2 (oracle/_ref), or None if it was never built."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libref_em2.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        R.ref_em2.argtypes = [dp, dp, dp, C.c_double, C.c_int]
        R.ref_em2_batch.argtypes = [C.c_size_t, dp, dp, dp]
        _REF = R
    return _REF


def use_reference_em2(on):
    """all_pairs(indep_geno=False) on the reference's own em2() (oracle/_ref, emOptim2.cpp as it lies) instead of the
    restated one; returns False if oracle/_ref was never built.  Process-wide; switch it back off after use."""
    L = lib()
    L.ngo_set_em2_hook.argtypes = [C.c_void_p]
    if not on:
        L.ngo_set_em2_hook(None)
        return True
    R = ref_lib()
    if R is None or not hasattr(R, "ref_em2_tls"):
        return False
    L.ngo_set_em2_hook(C.cast(R.ref_em2_tls, C.c_void_p))
    return True


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64)) if a is not None else None


class Taus:
    """gsl_rng_taus restatement (see ngsdist_oracle.c)."""

    def __init__(self, seed):
        self._st = (C.c_uint32 * 3)()
        lib().ngo_taus_set(self._st, int(seed) & 0xFFFFFFFFFFFFFFFF)

    def get(self):
        return lib().ngo_taus_get(self._st)

    def uniform(self):
        return lib().ngo_taus_uniform(self._st)

    def block_map(self, n_blocks):
        m = np.empty(n_blocks, dtype=np.uint64)
        lib().ngo_boot_block_map(self._st, n_blocks, _u64p(m))
        return m


# --------------------------------------------------------------------------
# loading + prep -> p[n_ind][n_sites][3] normal space (what gen_dist reads)
# --------------------------------------------------------------------------
def prep_binary(raw_site_major, n_ind, n_sites, in_logscale=False, call_geno=False,
                N_thresh=0.0, call_thresh=0.0):
    """raw_site_major: float64 array of n_sites*n_ind*3 as in the binary file."""
    raw = np.ascontiguousarray(raw_site_major, dtype=np.float64).reshape(-1)
    assert raw.size == n_ind * n_sites * 3
    out = np.empty_like(raw)
    bad = lib().ngo_prep_binary(_dp(raw), n_ind * n_sites, int(in_logscale), int(call_geno),
                                N_thresh, call_thresh, _dp(out))
    if bad:
        raise ValueError("NaN found! Is the file format correct?")  # read_data.cpp:42-45
    return np.ascontiguousarray(out.reshape(n_sites, n_ind, 3).transpose(1, 0, 2))


def _split_doubles(line):
    """split(buf, " \\t", &t): fields that are not entirely numeric are dropped
    (gen_func.cpp:390-417)."""
    out = []
    for tok in line.replace("\t", " ").split(" "):
        if tok == "":
            continue
        try:
            out.append(float(tok))
        except ValueError:
            pass
    return out


def load_text(path, n_ind, n_sites, in_probs, in_logscale=False, call_geno=False,
              N_thresh=0.0, call_thresh=0.0):
    """gz text input, read_data.cpp:48-103 (+ prep of ngsDist.cpp:165-174)."""
    L = lib()
    n_geno = 3 if in_probs else 1
    p = np.empty((n_ind, n_sites, 3), dtype=np.float64)
    tmp = (C.c_double * 3)()
    out = (C.c_double * 3)()
    L.ngo_prep_empty_line_one(int(call_geno), N_thresh, call_thresh, out)
    p[:, :, :] = np.array(out[:])
    opener = gzip.open if path.endswith(".gz") else open
    s = 0
    with opener(path, "rt") as fh:
        for line in fh:
            if s >= n_sites:
                raise ValueError("GENO file not at EOF. Check GENO file and number of sites!")
            line = line.rstrip("\n").rstrip("\r")
            if len(line) == 0:
                s += 1
                continue
            t = _split_doubles(line)
            if len(t) == 0 or (s == 0 and len(t) < n_ind * n_geno):
                continue  # header
            if len(t) < n_ind * n_geno:
                raise ValueError("wrong GENO file format. Less fields than expected!")
            ptr = t[len(t) - n_ind * n_geno:]
            for i in range(n_ind):
                if in_probs:
                    tmp[0], tmp[1], tmp[2] = ptr[3 * i], ptr[3 * i + 1], ptr[3 * i + 2]
                    with np.errstate(all="ignore"):
                        L.ngo_prep_text_probs_one(tmp, int(in_logscale), int(call_geno), N_thresh,
                                                  call_thresh, out)
                else:
                    if L.ngo_prep_text_geno_one(ptr[i], out):
                        raise ValueError("wrong GENO file format. Genotypes must be coded as {-1,0,1,2} !")
                p[i, s, 0], p[i, s, 1], p[i, s, 2] = out[0], out[1], out[2]
            s += 1
    if s < n_sites:
        raise ValueError("GENO file at premature EOF. Check GENO file and number of sites!")
    return p


# --------------------------------------------------------------------------
# hot path
# --------------------------------------------------------------------------
def n_pairs(n_ind):
    return n_ind * (n_ind - 1) // 2


def all_pairs(p, score=None, pairwise_del=False, indep_geno=True, site_src=None, n_sites=None,
              n_threads=1, want_iters=False):
    """p: [n_ind][n_sites_total][3] float64.  Returns (sum, cnt[, iters]) in
    row-major upper-triangle pair order (ngsDist.cpp:244-245)."""
    p = np.ascontiguousarray(p, dtype=np.float64)
    n_ind, n_tot, _ = p.shape
    if n_sites is None:
        n_sites = n_tot if site_src is None else len(site_src)
    score = DEFAULT_SCORE if score is None else np.ascontiguousarray(score, dtype=np.float64).reshape(9)
    npairs = n_pairs(n_ind)
    s = np.zeros(npairs, dtype=np.float64)
    c = np.zeros(npairs, dtype=np.uint64)
    it = np.zeros(npairs, dtype=np.uint64) if want_iters else None
    if site_src is not None:
        site_src = np.ascontiguousarray(site_src, dtype=np.uint64)
    rc = lib().ngo_all_pairs(_dp(p), n_ind, n_tot, _u64p(site_src), n_sites, _dp(score),
                             int(pairwise_del), int(indep_geno), int(n_threads), _dp(s), _u64p(c),
                             _u64p(it))
    assert rc == 0
    return (s, c, it) if want_iters else (s, c)


def em2(gl1, gl2, tole=0.001, max_iter=50):
    sfs = np.full(9, 1.0 / 9)
    g1 = np.ascontiguousarray(gl1, dtype=np.float64)
    g2 = np.ascontiguousarray(gl2, dtype=np.float64)
    n = lib().ngo_em2(_dp(sfs), _dp(g1), _dp(g2), tole, max_iter)
    return sfs, n


def ref_em2(gl1, gl2, tole=0.001, max_iter=50):
    R = ref_lib()
    sfs = np.full(9, 1.0 / 9)
    g1 = np.ascontiguousarray(gl1, dtype=np.float64)
    g2 = np.ascontiguousarray(gl2, dtype=np.float64)
    R.ref_em2(_dp(sfs), _dp(g1), _dp(g2), tole, max_iter)
    return sfs


def finish(sum_, cnt, tot_sites=0, evol_model=1):
    out = np.empty(len(sum_), dtype=np.float64)
    err = C.c_int(0)
    L = lib()
    for k in range(len(sum_)):
        out[k] = L.ngo_finish(float(sum_[k]), int(cnt[k]), int(tot_sites), int(evol_model), C.byref(err))
        if err.value:
            raise ValueError("invalid evolutionary model specified!")
    return out


def fmt_cell(v):
    buf = C.create_string_buffer(64)
    lib().ngo_format_cell(float(v), buf, 64)
    return buf.value.decode()


def format_matrix(dist_pairs, labels, prev_matrix=None):
    """The print block ngsDist.cpp:282-287.  Returns (text, full matrix)."""
    n = len(labels)
    m = np.zeros((n, n)) if prev_matrix is None else prev_matrix
    k = 0
    for i in range(n):
        for j in range(i + 1, n):
            m[i, j] = m[j, i] = dist_pairs[k]
            k += 1
    lines = ["", str(n)]
    for i in range(n):
        lines.append(labels[i] + "\t" + "\t".join(fmt_cell(v) for v in m[i]))
    return "\n".join(lines) + "\n", m


def default_labels(n_ind):
    return ["Ind_%d" % i for i in range(n_ind)]  # ngsDist.cpp:118-124


def boot_site_src(block_map, block_size):
    """site_src[block*B + s] = map[block]*B + s, ngsDist.cpp:426-434."""
    bm = np.asarray(block_map, dtype=np.uint64)
    return (bm[:, None] * np.uint64(block_size) + np.arange(block_size, dtype=np.uint64)[None, :]).reshape(-1)


def run_reference_flow(p, labels=None, score=None, pairwise_del=False, indep_geno=True, tot_sites=0,
                       evol_model=1, n_boot_rep=0, boot_block_size=1, seed=12345, n_threads=1,
                       raw=False):
    """The replicate loop of main(), ngsDist.cpp:217-289.  Returns the .dist text
    (and, if raw, the per-replicate (sum, cnt, dist) arrays)."""
    n_ind, n_sites, _ = p.shape
    labels = default_labels(n_ind) if labels is None else labels
    rng = Taus(seed)
    text = ""
    raws = []
    for rep in range(n_boot_rep + 1):
        site_src = None
        if rep > 0:
            n_sites -= n_sites % boot_block_size
            bm = rng.block_map(n_sites // boot_block_size)
            site_src = boot_site_src(bm, boot_block_size)
        s, c = all_pairs(p, score, pairwise_del, indep_geno, site_src, n_sites, n_threads)
        with np.errstate(all="ignore"):
            d = finish(s, c, tot_sites, evol_model)
        t, _ = format_matrix(d, labels)
        text += t
        raws.append((s, c, d))
    return (text, raws) if raw else text


def synth_indmajor(seed, n_ind, n_sites, miss_frac=0.0, i0=0, n_sub=None, s0=0):
    n_sub = n_ind if n_sub is None else n_sub
    p = np.empty((n_sub, n_sites, 3), dtype=np.float64)
    lib().ngo_synth_fill_indmajor(seed, n_ind, i0, n_sub, s0, n_sites, miss_frac, _dp(p))
    return p
