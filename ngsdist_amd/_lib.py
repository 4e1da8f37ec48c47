"""Loads libngsdist_amd.so (the HIP engine behind include/ngsdist_amd.h).

There is no fallback: if the library is missing or will not load, importing
fails loudly.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C ngsdist_amd/csrc`.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# NGSDIST_AMD_LIB: an A/B build of the same engine (tools/build_variant.sh) for the measuring tools; never a fallback
LIB_PATH = os.environ.get("NGSDIST_AMD_LIB") or os.path.join(_HERE, "libngsdist_amd.so")


class NgdConfig(C.Structure):
    _fields_ = [
        ("n_ind", C.c_uint64),
        ("n_sites", C.c_uint64),
        ("score", C.c_double * 9),
        ("pairwise_del", C.c_int32),
        ("indep_geno", C.c_int32),
        ("device", C.c_int32),
        ("kernel", C.c_int32),
        ("shard_rank", C.c_uint32),
        ("shard_world", C.c_uint32),
        ("variant", C.c_uint32),
        ("n_slices", C.c_uint32),
        ("wg_target", C.c_uint32),
        ("exact_shapes", C.c_uint32),
        ("single_image", C.c_uint32),
        ("second_image_mib", C.c_uint32),
    ]


class NgdPrep(C.Structure):
    _fields_ = [("in_logscale", C.c_int32), ("call_geno", C.c_int32), ("N_thresh", C.c_double),
                ("call_thresh", C.c_double)]


class NgdTiming(C.Structure):
    _fields_ = [
        ("ms_total", C.c_double),
        ("ms_accum", C.c_double),
        ("ms_reduce", C.c_double),
        ("ms_count", C.c_double),
        ("pair_sites", C.c_uint64),
        ("launches", C.c_uint64),
    ]


class NgdSpillTiming(C.Structure):
    _fields_ = [("ms_weights", C.c_double), ("ms_terms", C.c_double), ("ms_sanitize", C.c_double), ("ms_contract", C.c_double)] + [
        (k, C.c_uint64) for k in ("chunks", "sites", "unit_sites", "units", "slot_groups", "slot_groups_live", "matrices",
                                  "matrix_groups", "contract_launches")]


class NgdFixupInfo(C.Structure):
    _fields_ = [("flagged", C.c_uint64), ("recomputed", C.c_uint64), ("skipped", C.c_uint64), ("ms", C.c_double),
                ("by_pass", C.c_uint64)]


# every symbol include/ngsdist_amd.h declares (tests/test_abi.py checks the header against this)
EXPORTS = [
    "ngd_last_error", "ngd_abi_version", "ngd_device_count", "ngd_create", "ngd_destroy",
    "ngd_upload_sites", "ngd_upload_ind_major", "ngd_commit", "ngd_stage_acquire", "ngd_stage_submit",
    "ngd_upload_raw_sites", "ngd_synth_fill", "ngd_synth_fill_range", "ngd_run", "ngd_run_mult", "ngd_run_mult_device",
    "ngd_run_device", "ngd_run_batch", "ngd_run_batch_device", "ngd_run_mult_batch",
    "ngd_run_mult_batch_device", "ngd_run_job", "ngd_run_job_device", "ngd_run_job_dist", "ngd_run_batch_dist", "ngd_run_mult_batch_dist", "ngd_fetch_matrix", "ngd_drop_caches", "ngd_set_option", "ngd_last_timing", "ngd_last_spill_timing", "ngd_last_fixup", "ngd_image_mode", "ngd_last_shader_clock", "ngd_last_em_work", "ngd_finish", "ngd_finish_stream", "ngd_format_matrix", "ngd_taus_seed", "ngd_taus_get",
    "ngd_taus_uniform", "ngd_boot_block_map", "ngd_n_pairs", "ngd_pair_index", "ngd_device_bytes", "ngd_device_memory", "ngd_shard_of_pair", "ngd_shard_map",
    "ngd_score_congruence",
]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "ngsdist_amd: %s is missing -- the HIP engine was not built (run __graft_entry__.build() "
            "or `make -C ngsdist_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: if PyTorch is present, let ITS libamdhip64.so.7 be
    # the one already mapped when our DT_NEEDED entry of the same soname is resolved,
    # so device pointers / streams can be shared with torch (RCCL gather in bench.py).
    if "torch" not in sys.modules and not os.environ.get("NGD_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    try:
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as exc:
        raise ImportError("ngsdist_amd: cannot load %s: %s (no CPU fallback)" % (LIB_PATH, exc))
    vp, dp, u64, u64p = C.c_void_p, C.POINTER(C.c_double), C.c_uint64, C.POINTER(C.c_uint64)
    u32p = C.POINTER(C.c_uint32)
    L.ngd_last_error.restype = C.c_char_p
    L.ngd_abi_version.restype = C.c_int
    L.ngd_device_count.restype = C.c_int
    L.ngd_create.argtypes = [C.POINTER(NgdConfig), C.POINTER(vp)]
    L.ngd_destroy.argtypes = [vp]
    L.ngd_destroy.restype = None
    L.ngd_upload_sites.argtypes = [vp, dp, u64, u64]
    L.ngd_upload_ind_major.argtypes = [vp, dp]
    L.ngd_commit.argtypes = [vp]
    L.ngd_stage_acquire.argtypes = [vp, C.POINTER(dp), u64p]
    L.ngd_stage_submit.argtypes = [vp, u64, u64, C.POINTER(NgdPrep)]
    L.ngd_upload_raw_sites.argtypes = [vp, dp, u64, u64, C.POINTER(NgdPrep)]
    L.ngd_synth_fill.argtypes = [vp, u64, C.c_double]
    L.ngd_synth_fill_range.argtypes = [vp, u64, C.c_double, u64]
    L.ngd_run_mult.argtypes = [vp, u32p, u64, u64, dp, u64p]
    L.ngd_run_mult_device.argtypes = [vp, u32p, u64, u64, vp, vp]
    L.ngd_run.argtypes = [vp, u64p, u64, u64, dp, u64p]
    L.ngd_run_device.argtypes = [vp, u64p, u64, u64, vp, vp]
    L.ngd_run_batch.argtypes = [vp, u64p, C.c_uint32, u64, u64, dp, u64p]
    L.ngd_run_batch_device.argtypes = [vp, u64p, C.c_uint32, u64, u64, vp, vp]
    L.ngd_run_mult_batch.argtypes = [vp, u32p, C.c_uint32, u64, u64, dp, u64p]
    L.ngd_run_job.argtypes = [vp, u64p, C.c_uint32, u64, u64, dp, u64p]
    L.ngd_run_job_device.argtypes = [vp, u64p, C.c_uint32, u64, u64, vp, vp]
    L.ngd_fetch_matrix.argtypes = [vp, C.c_uint32, dp, u64p]
    L.ngd_run_job_dist.argtypes = [vp, u64p, C.c_uint32, u64, u64, u64, u64, dp]
    L.ngd_run_batch_dist.argtypes = [vp, u64p, C.c_uint32, u64, u64, u64, u64, dp]
    L.ngd_run_mult_batch_dist.argtypes = [vp, u32p, C.c_uint32, u64, u64, u64, u64, dp]
    L.ngd_run_mult_batch_device.argtypes = [vp, u32p, C.c_uint32, u64, u64, vp, vp]
    L.ngd_drop_caches.argtypes = [vp]
    L.ngd_set_option.argtypes = [vp, C.c_int, u64]
    L.ngd_last_timing.argtypes = [vp, C.POINTER(NgdTiming)]
    L.ngd_last_spill_timing.argtypes = [vp, C.POINTER(NgdSpillTiming)]
    L.ngd_last_fixup.argtypes = [vp, C.POINTER(NgdFixupInfo)]
    L.ngd_image_mode.argtypes = [vp, C.POINTER(C.c_int)]
    L.ngd_image_mode.restype = C.c_int
    L.ngd_last_em_work.argtypes = [vp, u64p, u64p]
    L.ngd_last_shader_clock.argtypes = [vp, dp]
    L.ngd_finish.argtypes = [dp, u64p, u64, u64, u64, dp]
    L.ngd_finish_stream.argtypes = [dp, u64p, u64, u64, u64, dp, u64p]
    L.ngd_format_matrix.argtypes = [dp, u64, C.POINTER(C.c_char_p), C.c_char_p, u64, C.c_uint32]
    L.ngd_format_matrix.restype = C.c_int64
    L.ngd_device_memory.argtypes = [C.c_int, u64p, u64p]
    L.ngd_taus_seed.argtypes = [u32p, u64]
    L.ngd_taus_seed.restype = None
    L.ngd_taus_get.argtypes = [u32p]
    L.ngd_taus_get.restype = C.c_uint32
    L.ngd_taus_uniform.argtypes = [u32p]
    L.ngd_taus_uniform.restype = C.c_double
    L.ngd_boot_block_map.argtypes = [u32p, u64, u64p]
    L.ngd_boot_block_map.restype = None
    L.ngd_n_pairs.argtypes = [u64]
    L.ngd_n_pairs.restype = u64
    L.ngd_pair_index.argtypes = [u64, u64, u64]
    L.ngd_pair_index.restype = u64
    L.ngd_shard_of_pair.argtypes = [u64, u64, u64, C.c_uint32]
    L.ngd_shard_of_pair.restype = C.c_uint32
    L.ngd_shard_map.argtypes = [u64, C.c_uint32, C.POINTER(C.c_int32)]
    L.ngd_shard_map.restype = None
    L.ngd_score_congruence.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ngd_score_congruence.restype = C.c_int
    L.ngd_device_bytes.argtypes = [vp]
    L.ngd_device_bytes.restype = u64
    _lib = L
    return L
