"""ctypes loader for libknoxgpu.so (the C ABI declared in include/knoxgpu.h).

The product path has no CPU fallback: if the HIP library was not built this raises, and every
entry point returns a knox_err when no MI355X is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KNOXGPU_SO: diagnostics builds only (knoxdb_amd/diag/, `make diag`); the product is the in-tree library
SO_PATH = os.environ.get("KNOXGPU_SO") or os.path.join(_HERE, "libknoxgpu.so")

COMM_ID_BYTES = 128


class KnoxError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = "knox error %d" % code
        try:
            msg += " (%s)" % lib().knox_strerror(code).decode()
        except Exception:  # pragma: no cover
            pass
        super().__init__(msg + (": " + what if what else ""))


class Pred(C.Structure):
    """knox_pred (include/knoxgpu.h) -- one postfix node of a filter.Node tree."""
    _fields_ = [("kind", C.c_int32), ("nchild", C.c_int32), ("col", C.c_int32), ("op", C.c_int32),
                ("a", C.c_uint64), ("b", C.c_uint64), ("val", C.c_char_p), ("val_len", C.c_uint64),
                ("val2", C.c_char_p), ("val2_len", C.c_uint64)]


class Agg(C.Structure):
    _fields_ = [("func", C.c_int32), ("col", C.c_int32)]


class AggState(C.Structure):
    """knox_agg_state -- mergeable partial aggregate of one column."""
    _fields_ = [("dtype", C.c_int32), ("flags", C.c_uint32), ("count", C.c_int64), ("isum", C.c_uint64),
                ("fsum_hi", C.c_double), ("fsum_lo", C.c_double),
                ("min_bits", C.c_uint64), ("max_bits", C.c_uint64),
                ("first_bits", C.c_uint64), ("last_bits", C.c_uint64),
                ("min_row", C.c_int64), ("max_row", C.c_int64),
                ("first_row", C.c_int64), ("last_row", C.c_int64),
                ("m_shift", C.c_double), ("m_dev_hi", C.c_double), ("m_dev_lo", C.c_double),
                ("m_sq_hi", C.c_double), ("m_sq_lo", C.c_double)]


class ArenaInfo(C.Structure):
    """knox_arena_info -- the context's device-memory arena."""
    _fields_ = [("budget_bytes", C.c_uint64), ("mapped_bytes", C.c_uint64), ("used_bytes", C.c_uint64), ("peak_used_bytes", C.c_uint64),
                ("spare_write_buffer_bytes", C.c_uint64), ("mixed_zone_bytes", C.c_uint64), ("write_buffers", C.c_int32),
                ("chunks", C.c_int32), ("probes", C.c_int32), ("enabled", C.c_int32), ("probe_launches", C.c_int32), ("probe_ms", C.c_float), ("unprobed_chunks", C.c_int32), ("unverified_write_buffers", C.c_int32)]


class ScanOut(C.Structure):
    _fields_ = [("count", C.c_int64), ("rows_scanned", C.c_int64)]


# name -> (restype, argtypes); every symbol include/knoxgpu.h declares
_vp, _i, _i64, _u64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_size_t
_pp = C.POINTER(C.c_void_p)
_pi = C.POINTER(C.c_int)
SIGNATURES = {
    "knox_abi_version": (_i, []),
    "knox_strerror": (C.c_char_p, [_i]),
    "knox_device_count": (_i, [_pi]),
    "knox_last_hip_error": (_i, [_vp]),
    "knox_ctx_open": (_i, [_i, _pp]),
    "knox_ctx_open_stream": (_i, [_i, _vp, _pp]),
    "knox_ctx_close": (_i, [_vp]),
    "knox_ctx_sync": (_i, [_vp]),
    "knox_ctx_stream": (_vp, [_vp]),
    "knox_timer_start": (_i, [_vp]),
    "knox_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "knox_ctx_set_tuning": (_i, [_vp, _i]),
    "knox_prof_enable": (_i, [_vp, _i]),
    "knox_prof_read": (_i, [_vp, C.POINTER(C.c_double), _pi]),
    "knox_prof_read_each": (_i, [_vp, C.POINTER(C.c_float), _i, _pi]),
    "knox_scan_occupancy_mismatches": (_i, []),
    "knox_scan_launch_shape": (_i, [_i, _i, _i, _i, _i, _pi, C.POINTER(C.c_uint32)]),
    "knox_cmp": (_i64, [_vp, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "knox_col_upload": (_i, [_vp, _i, _vp, _sz, _sz, _pp]),
    "knox_col_create": (_i, [_vp, _i, _sz, _sz, _pp]),
    "knox_col_append_async": (_i, [_vp, _vp, _vp, _sz]),
    "knox_col_reset": (_i, [_vp, _vp]),
    "knox_col_upload_async": (_i, [_vp, _i, _vp, _sz, _sz, _pp]),
    "knox_col_wait": (_i, [_vp, _vp]),
    "knox_col_ready": (_i, [_vp]),
    "knox_col_upload_bytes": (_i, [_vp, _vp, _sz, _vp, _sz, _sz, _pp]),
    "knox_col_bytes_create": (_i, [_vp, _sz, _sz, _sz, _pp]),
    "knox_col_bytes_append_pack": (_i, [_vp, _vp, _vp, _sz, _vp, _sz]),
    "knox_col_bytes_download_pack": (_i, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz), _vp, _sz, C.POINTER(_sz)]),
    "knox_col_pool_bytes": (_sz, [_vp]),
    "knox_col_generate_bytes": (_i, [_vp, _u64, _sz, _sz, C.c_uint32, C.c_uint32, C.c_uint32, _pp]),
    "knox_packed_upload": (_i, [_vp, _i, _vp, _vp, _sz, _sz, _pp]),
    "knox_packed_create": (_i, [_vp, _i, _sz, _sz, _sz, _pp]),
    "knox_packed_append_async": (_i, [_vp, _vp, _vp, _vp, _sz]),
    "knox_packed_reset": (_i, [_vp, _vp]),
    "knox_packed_wait": (_i, [_vp, _vp]),
    "knox_packed_free": (_i, [_vp]),
    "knox_packed_rows": (_sz, [_vp]),
    "knox_packed_bytes": (_sz, [_vp]),
    "knox_packed_scan": (_i, [_vp, _vp, _i, _u64, _u64, _i, _vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(AggState)]),
    "knox_enc_upload": (_i, [_vp, _i, _vp, _vp, _sz, _sz, _pp]),
    "knox_enc_free": (_i, [_vp]),
    "knox_enc_rows": (_sz, [_vp]),
    "knox_enc_bytes": (_sz, [_vp]),
    "knox_enc_container_bytes": (_sz, [_vp]),
    "knox_enc_kinds": (_i, [_vp, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz)]),
    "knox_enc_scan": (_i, [_vp, _vp, _i, _u64, _u64, _i, _vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(AggState)]),
    "knox_col_upload_wide_containers": (_i, [_vp, _i, _vp, _vp, _sz, _sz, _pp]),
    "knox_bits_upload_bitmap_container": (_i, [_vp, _vp, C.c_char_p, _sz]),
    "knox_col_upload_string_containers": (_i, [_vp, _vp, _vp, _sz, _sz, _pp]),
    "knox_col_upload_raw_containers": (_i, [_vp, _i, _vp, _vp, _sz, _sz, _pp]),
    "knox_col_wrap": (_i, [_vp, _i, _vp, _sz, _sz, _pp]),
    "knox_col_generate": (_i, [_vp, _i, _i, _u64, _u64, _sz, _i64, _u64, _sz, _pp]),
    "knox_col_download": (_i, [_vp, _vp, _sz, _sz, _vp]),
    "knox_col_free": (_i, [_vp]),
    "knox_col_rows": (_sz, [_vp]),
    "knox_col_dtype": (_i, [_vp]),
    "knox_col_device_ptr": (_vp, [_vp]),
    "knox_bits_alloc": (_i, [_vp, _sz, _pp]),
    "knox_bits_alloc_for_cols": (_i, [_vp, _sz, _vp, _i, _pp]),
    "knox_bits_free": (_i, [_vp]),
    "knox_bits_size": (_sz, [_vp]),
    "knox_bits_device_ptr": (_vp, [_vp]),
    "knox_bits_upload": (_i, [_vp, _vp, _vp]),
    "knox_bits_download": (_i, [_vp, _vp, _vp]),
    "knox_bits_zero": (_i, [_vp, _vp]),
    "knox_bits_one": (_i, [_vp, _vp]),
    "knox_bits_copy": (_i, [_vp, _vp, _vp]),
    "knox_bits_and": (_i, [_vp, _vp, _vp]),
    "knox_bits_andnot": (_i, [_vp, _vp, _vp]),
    "knox_bits_or": (_i, [_vp, _vp, _vp]),
    "knox_bits_xor": (_i, [_vp, _vp, _vp]),
    "knox_bits_neg": (_i, [_vp, _vp]),
    "knox_bits_and_flag": (_i, [_vp, _vp, _vp, _pi, _pi]),
    "knox_bits_or_flag": (_i, [_vp, _vp, _vp, _pi, _pi]),
    "knox_bits_popcount": (_i, [_vp, _vp, C.POINTER(_i64)]),
    "knox_bits_popcount_segments": (_i, [_vp, _vp, _sz, _vp, _sz]),
    "knox_bits_match_bool": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "knox_bits_indexes": (_i, [_vp, _vp, _vp, _sz, C.POINTER(_sz)]),
    "knox_bits_indexes_view": (_i, [_vp, _vp, C.POINTER(_vp), C.POINTER(_sz)]),
    "knox_bits_and_flag_async": (_i, [_vp, _vp, _vp]),
    "knox_bits_or_flag_async": (_i, [_vp, _vp, _vp]),
    "knox_bits_popcount_async": (_i, [_vp, _vp]),
    "knox_bits_collect": (_i, [_vp, _pi, _pi, C.POINTER(_i64)]),
    "knox_scan": (_i, [_vp, C.POINTER(Pred), _i, _pp, _i, C.POINTER(Agg), _i, _u64, _vp, _vp, _vp,
                       C.POINTER(AggState), C.POINTER(ScanOut)]),
    "knox_scan_async": (_i, [_vp, C.POINTER(Pred), _i, _pp, _i, C.POINTER(Agg), _i, _u64, _vp, _vp, _i]),
    "knox_scan_collect": (_i, [_vp, _vp, C.POINTER(AggState), _i, C.POINTER(ScanOut)]),
    "knox_match_range_vectors": (_i, [_vp, _vp, _vp, _i, _u64, _u64, _vp, _vp]),
    "knox_match_range_vectors_set": (_i, [_vp, _vp, _vp, _i, _vp, _sz, _vp, _vp]),
    "knox_match_range_vectors_wide": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "knox_match_range_vectors_bytes": (_i, [_vp, _vp, _vp, _i, C.c_char_p, _u64, C.c_char_p, _u64, _vp, _vp]),
    "knox_reduce": (_i, [_vp, _vp, _vp, _u64, C.POINTER(AggState)]),
    "knox_host_alloc": (_i, [_vp, _sz, _pp]),
    "knox_host_free": (_i, [_vp, _vp]),
    "knox_ctx_set_placement": (_i, [_vp, _i]),
    "knox_ctx_placement_info": (_i, [_vp, C.POINTER(_i64), _pi]),
    "knox_ctx_set_arena": (_i, [_vp, _sz]),
    "knox_ctx_reserve": (_i, [_vp, _sz]),
    "knox_col_placement_class": (_i, [_vp, _i]),
    "knox_ctx_set_probe_budget": (_i, [_vp, _i, C.c_double]),
    "knox_ctx_arena_info": (_i, [_vp, C.POINTER(ArenaInfo)]),
    "knox_ctx_arena_classes": (_i, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "knox_ctx_arena_trim": (_i, [_vp, C.POINTER(_sz)]),
    "knox_bits_pair_check": (_i, [_vp, C.POINTER(C.c_double), _pi]),
    "knox_bits_wrap": (_i, [_vp, _vp, _sz, _pp]),
    "knox_bits_alloc_for": (_i, [_vp, _sz, _vp, _pp]),
    "knox_match_wide_set": (_i, [_vp, _pp, _i, _i, C.POINTER(_u64), _sz, _vp, _vp]),
    "knox_match_wide": (_i, [_vp, _pp, _i, _i, C.POINTER(_u64), C.POINTER(_u64), _vp, _vp, C.POINTER(_i64)]),
    "knox_reduce_moments": (_i, [_vp, _vp, _vp, C.POINTER(AggState)]),
    "knox_reduce_buckets": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _sz, _u64, _i, C.POINTER(AggState)]),
    "knox_reduce_buckets_view": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _sz, _u64, _i, C.POINTER(C.POINTER(AggState))]),
    "knox_reduce_buckets_values": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _sz, _i, _i, C.POINTER(_vp), C.POINTER(_vp)]),
    "knox_match_regexp": (_i, [_vp, _vp, _vp, _u64, _vp, _vp, C.POINTER(_i64)]),
    "knox_regexp_match_value": (_i, [_vp, _u64, _vp, _u64]),
    "knox_regexp_unicode_version": (C.c_char_p, []),
    "knox_reduce_buckets_sequential": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _sz, _i, _i, _vp, _vp]),
    "knox_reduce_sequential": (_i, [_vp, _vp, _vp, _i, _vp, _pi, C.POINTER(_i64)]),
    "knox_agg_value": (_i, [C.POINTER(AggState), _i, _vp, _pi, C.POINTER(_i64)]),
    "knox_fill_windows": (_i, [_i, _i, _vp, _vp, _sz, _i64, _i64, _i64, _sz, _vp, _vp, C.POINTER(_sz)]),
    "knox_agg_combine": (_i, [C.POINTER(AggState), _i, C.POINTER(AggState)]),
    "knox_bloom_upload": (_i, [_vp, _vp, _vp, _sz, _pp]),
    "knox_bloom_build": (_i, [_vp, _vp, _i, _pp]),
    "knox_bloom_build_u64": (_i, [_vp, _vp, _i, _pp]),
    "knox_bloom_packs": (_sz, [_vp]),
    "knox_bloom_build_info": (_i, [_vp, _pi, _pi]),
    "knox_windows_last_path": (_i, [_vp]),
    "knox_bloom_bytes": (_sz, [_vp]),
    "knox_bloom_download": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "knox_bloom_probe": (_i, [_vp, _vp, _u64, _vp]),
    "knox_bloom_free": (_i, [_vp]),
    "knox_hash_bytes": (_u64, [C.c_char_p, _sz]),
    "knox_hash_value": (_u64, [_i, _vp]),
    "knox_comm_unique_id": (_i, [_vp]),
    "knox_comm_init": (_i, [_vp, _i, _i, _vp]),
    "knox_comm_allgather_states": (_i, [_vp, C.POINTER(AggState), _i, C.POINTER(AggState), C.POINTER(_i64)]),
    "knox_comm_allreduce_sum_count": (_i, [_vp, C.POINTER(_u64), C.POINTER(_i64)]),
    "knox_comm_allreduce_sum_count_start": (_i, [_vp, _u64, _i64]),
    "knox_comm_allreduce_sum_count_finish": (_i, [_vp, C.POINTER(_u64), C.POINTER(_i64)]),
    "knox_comm_ranks": (_i, [_vp, _pi, _pi]),
    "knox_comm_destroy": (_i, [_vp]),
}

_lib = None


def lib():
    """Load libknoxgpu.so (built by __graft_entry__.build() / `make -C knoxdb_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError("libknoxgpu.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                              "there is no CPU fallback for the GPU path" % SO_PATH)
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise KnoxError(rc, what)
    return rc
