"""ctypes binding of libb3m.so (include/b3m.h).  Fails loudly when the library is missing."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# B3M_LIB: another build of the same library (the LDS-checking debug build, `make debug`); never a different backend
_SO = os.environ.get("B3M_LIB") or os.path.join(_HERE, "libb3m.so")

INPUT_TYPES = {"bytestream": 0, "compactstream": 1, "pac": 2, "pacterm": 3, "lz4": 4, "utf-8": 5}
SYMBOLS32 = 6      # B3M_INPUT_SYMBOLS32: plans only (n uint32 symbols below 2^21), not a file type of the reference


class LibraryMissing(RuntimeError):
    pass


class B3MError(RuntimeError):
    """error convention of the reference (libmaus2::exception::LibMausException,
    /root/reference/src/bwtb3m.cpp:34,67): an exception carrying the message."""


class Options(ctypes.Structure):
    _fields_ = [
        ("input_fn", ctypes.c_char_p), ("inputtype", ctypes.c_char_p), ("output_fn", ctypes.c_char_p),
        ("tmpprefix", ctypes.c_char_p), ("sparsetmpprefix", ctypes.c_char_p),
        ("mem", ctypes.c_uint64), ("sasamplingrate", ctypes.c_uint64), ("isasamplingrate", ctypes.c_uint64),
        ("largelcpthres", ctypes.c_uint64),
        ("numthreads", ctypes.c_uint32), ("bwtonly", ctypes.c_uint32), ("copyinputtomemory", ctypes.c_uint32),
        ("verbose", ctypes.c_uint32), ("num_gpus", ctypes.c_int32), ("blocksize", ctypes.c_uint64),
    ]


class Result(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_uint64), ("num_blocks", ctypes.c_uint64), ("levels", ctypes.c_uint64), ("p0", ctypes.c_uint64),
        ("num_samples", ctypes.c_uint64), ("tie_rounds", ctypes.c_uint64), ("tie_elements", ctypes.c_uint64),
        ("ms_total", ctypes.c_double), ("ms_sort", ctypes.c_double), ("ms_gap", ctypes.c_double),
        ("ms_interleave", ctypes.c_double),
        ("slice_offset", ctypes.c_uint64), ("slice_len", ctypes.c_uint64), ("has_p0", ctypes.c_uint64),
        ("tie_overhang", ctypes.c_uint64), ("tie_slow", ctypes.c_uint64),
    ]


class E2ETimes(ctypes.Structure):
    _fields_ = [("h2d_ms", ctypes.c_double), ("construct_ms", ctypes.c_double), ("d2h_tail_ms", ctypes.c_double),
                ("total_ms", ctypes.c_double), ("streamed_bytes", ctypes.c_uint64)]


class KernelTime(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("ms", ctypes.c_double), ("launches", ctypes.c_uint64),
                ("alg_bytes", ctypes.c_uint64)]


# every symbol include/b3m.h declares: (name, restype, argtypes)
_u64, _p, _i, _cp = ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p
SYMBOLS = [
    ("b3m_version", _i, []),
    ("b3m_last_error", _cp, []),
    ("b3m_switch_name", _cp, [_i]),
    ("b3m_switch_doc", _cp, [_i]),
    ("b3m_parse_input_type", _i, [_cp]),
    ("b3m_input_type_name", _cp, [_i]),
    ("b3m_options_init", None, [ctypes.POINTER(Options)]),
    ("b3m_compute_bwt", _i, [ctypes.POINTER(Options), _i]),
    ("b3m_plan_sampling_rate", _u64, [_u64, _i, _u64]),
    ("b3m_plan_create", _p, [_u64, _i, _u64, _u64, _u64]),
    ("b3m_plan_create_sharded", _p, [_u64, _i, _i, _i, _u64, _u64]),
    ("b3m_plan_slice_capacity", _u64, [_p]),
    ("b3m_plan_destroy", None, [_p]),
    ("b3m_plan_num_blocks", _u64, [_p]),
    ("b3m_plan_host_tier", _i, [_p]),
    ("b3m_plan_warning", _cp, [_p]),
    ("b3m_plan_workspace_bytes", _u64, [_p]),
    ("b3m_plan_text_bytes", _u64, [_p]),
    ("b3m_plan_text_buffer", _p, [_p]),
    ("b3m_plan_run", _i, [_p, _p, _p, _p]),
    ("b3m_plan_run_host", _i, [_p, _p, _p, ctypes.POINTER(E2ETimes), _p]),
    ("b3m_plan_set_packed_output", _i, [_p, _i]),
    ("b3m_host_alloc", _p, [_u64]),
    ("b3m_host_free", None, [_p]),
    ("b3m_compact_info", _i, [_p, _u64, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(_u64), ctypes.POINTER(_u64)]),
    ("b3m_compact_decode", _i, [_p, _u64, ctypes.c_uint32, _p, _p]),
    ("b3m_plan_set_keep_sa", _i, [_p, _i]),
    ("b3m_plan_has_sa", _i, [_p]),
    ("b3m_plan_tie_pending", _i, [_p, ctypes.POINTER(_u64)]),
    ("b3m_plan_tie_export", _i, [_p, _p, _p, _p]),
    ("b3m_plan_tie_finish", _i, [_p, _p, _p, _u64, _u64, _p, _p]),
    ("b3m_plan_set_tie_mode", _i, [_p, _i]),
    ("b3m_plan_list_counts", _i, [_p, ctypes.POINTER(_u64), ctypes.POINTER(_u64), ctypes.POINTER(_u64), ctypes.POINTER(_i)]),
    ("b3m_plan_list_fits", _i, [_p, _u64, _u64]),
    ("b3m_plan_list_export", _i, [_p, _p, _p, _p]),
    ("b3m_plan_list_finish", _i, [_p, _p, _u64, _p, _u64, _u64, _u64, _i, _p, _p]),
    ("b3m_plan_result", _i, [_p, ctypes.POINTER(Result)]),
    ("b3m_plan_samples", _i, [_p, _p, _u64]),
    ("b3m_plan_histogram", _i, [_p, _p]),
    ("b3m_plan_lf_walk", _i, [_p, _p, _u64, _p, _p, _p]),
    ("b3m_plan_lf_walk_dev", _i, [_p, _p, _u64, _p, _p, _p]),
    ("b3m_plan_samples_raw", _i, [_p, ctypes.POINTER(_p), ctypes.POINTER(_u64)]),
    ("b3m_plan_samples_device", _i, [_p, ctypes.POINTER(_p), ctypes.POINTER(_u64), ctypes.POINTER(_u64), _p]),
    ("b3m_lf_walk", _i, [_p, _u64, _i, _p, _u64, _p, _p, _u64, _u64, _p, _p, _p, _p]),
    ("b3m_check_bwt", _i, [_cp, _cp, _cp, _i, _p]),
    ("b3m_lf_walk_job", _i, [_p, _p, _p]),
    ("b3m_lf_walk_chains", _i, [_p, _u64, _u64, _p, _p]),
    ("b3m_merge_samples", _i, [_p, _p, _u64, _p]),
    ("b3m_merge_sample_pairs", _i, [_p, _p, _u64, _p, _p]),
    ("b3m_compute_ssa", _i, [_cp, _u64, _u64, _i]),
    ("b3m_plan_kernel_times", _i, [_p, ctypes.POINTER(KernelTime), _i]),
    ("b3m_histogram", _i, [_p, _u64, _i, _u64, _u64, _p, _p]),
    ("b3m_block_sort", _i, [_p, _u64, _i, _u64, _u64, _p, _p, _p, ctypes.POINTER(_u64), _p]),
    ("b3m_block_sort_abs", _i, [_p, _u64, _i, _u64, _u64, _p, _p, _p, ctypes.POINTER(_u64), _p]),
    ("b3m_rank_of", _i, [_p, _u64, _i, _p, _u64, _p, _u64, _p, _p]),
    ("b3m_gap", _i, [_p, _u64, _i, _u64, _u64, _u64, _p, _u64, _p, _p, _u64, _u64, _p, _p, _p, _p]),
    ("b3m_gap_chains", _i, [_p, _u64, _i, _u64, _u64, _u64, _p, _u64, _p, _p, _u64, _u64, _u64, _p, ctypes.POINTER(_u64), _p, _p]),
    ("b3m_sort_u32", _i, [_p, _u64, _i, _p, _p]),
    ("b3m_bits_place", _i, [_p, _u64, _u64, _p, _p]),
    ("b3m_interleave", _i, [_p, _p, _u64, _p, _u64, _p, _u64, ctypes.POINTER(_u64), _p]),
    ("b3m_dist_create", _p, [_u64, _i, _u64, _u64, _u64]),
    ("b3m_dist_destroy", None, [_p]),
    ("b3m_dist_workspace_bytes", _u64, [_p]),
    ("b3m_dist_set_text", _i, [_p, _p, _p]),
    ("b3m_dist_leaf", _i, [_p, _u64, _u64, _p, _p, ctypes.POINTER(_u64), _p, _p]),
    ("b3m_dist_leaf_hist", _i, [_p, _p]),
    ("b3m_dist_rank_of", _i, [_p, _p, _u64, _p, _p]),
    ("b3m_dist_rank_of_dev", _i, [_p, _p, _u64, _p, _p]),
    ("b3m_dist_endbit", _i, [_p, _p, _u64, _p, _u64, _p]),
    ("b3m_bits_copy", _i, [_p, _p, _u64, _u64, _p]),
    ("b3m_dist_pack2", _i, [_p, _p, _u64, _p, _p]),
    ("b3m_dist_build_dict", _i, [_p, _p, _p, _i, _u64, _u64, _p]),
    ("b3m_dist_walk", _i, [_p, _u64, _u64, _u64, _p, _p, _p, _p, _u64, _u64, _u64, _p, _i, _p, _p, ctypes.POINTER(_u64), _p]),
    ("b3m_dist_l_bytes", _i, [_p, _u64, _u64, _u64, _p, _p]),
    ("b3m_dist_r_bytes", _i, [_p, _p, _p, _i, _u64, _u64, _p, _p]),
    ("b3m_dist_sort_u32", _i, [_p, _p, _u64, _i, _p, _p]),
    ("b3m_dist_interleave", _i, [_p, _p, _p, _u64, _p, _u64, _p, _p]),
    ("b3m_rl_write", _i, [_cp, _p, _u64, _u64]),
    ("b3m_rl_read", _i, [_cp, _p, _u64, ctypes.POINTER(_u64)]),
    ("b3m_rl_write_device", _i, [_cp, _p, _u64, _u64, _p]),
    ("b3m_rl_read_symbols", _i, [_cp, _p, _u64, ctypes.POINTER(_u64)]),
    ("b3m_rl_length", _i, [_cp, ctypes.POINTER(_u64)]),
]

_lib = None


def so_path():
    return _SO


def lib():
    """the loaded library; raises LibraryMissing (no fallback) if it has not been built"""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise LibraryMissing(
                "%s not found: build it with `make` (or __graft_entry__.build()); bwtb3m_amd has no "
                "CPU fallback" % _SO)
        L = ctypes.CDLL(_SO)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)      # AttributeError if a declared symbol is not exported
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise B3MError(lib().b3m_last_error().decode("utf-8", "replace"))
