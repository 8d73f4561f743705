"""ctypes binding of libarrow_hip.so (include/arrow_hip.h).

The product path has NO CPU fallback: if the HIP library is missing, or no GPU is
visible when a Context is created, this raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AH_LIB_PATH") or os.path.join(_HERE, "lib", "libarrow_hip.so")  # AH_LIB_PATH: ablation builds

# status codes (include/arrow_hip.h)
AH_OK = 0
AH_INVALID_ARGUMENT = 1
AH_COMPUTE_ERROR = 2
AH_ARITHMETIC_OVERFLOW = 3
AH_DIVIDE_BY_ZERO = 4
AH_CAST_ERROR = 5
AH_OFFSET_OVERFLOW = 6
AH_NOT_YET_IMPLEMENTED = 7
AH_OFFSET_OVERFLOW_ERROR = 8
AH_C_DATA_INTERFACE = 9
AH_IPC_ERROR = 10
AH_PARSE_ERROR = 11
AH_PANIC = 100
AH_HIP_ERROR = 101
AH_OUT_OF_MEMORY = 102
AH_COMM_ERROR = 103

# physical types
AH_BOOL, AH_INT8, AH_INT16, AH_INT32, AH_INT64 = 1, 2, 3, 4, 5
AH_UINT8, AH_UINT16, AH_UINT32, AH_UINT64 = 6, 7, 8, 9
AH_FLOAT32, AH_FLOAT64, AH_FIXED16, AH_FIXED32 = 10, 11, 12, 13
AH_UTF8, AH_LARGE_UTF8, AH_FLOAT16 = 14, 15, 16
AH_UTF8_VIEW, AH_BINARY_VIEW = 17, 18

# logical ids of ah_data_type (the casts whose arithmetic depends on the logical type)
AH_DT_DATE32, AH_DT_DATE64, AH_DT_TIME32, AH_DT_TIME64, AH_DT_TIMESTAMP, AH_DT_DURATION = 32, 33, 34, 35, 36, 37
AH_DT_INTERVAL, AH_DT_DECIMAL128 = 38, 39

AH_OUT_BORROWED = 1


class DataTypeDesc(C.Structure):
    """ah_data_type / orc_data_type (identical layout)."""
    _fields_ = [("id", C.c_int32), ("unit", C.c_int32), ("has_tz", C.c_int32), ("tz_offset_seconds", C.c_int32),
                ("precision", C.c_int32), ("scale", C.c_int32)]


class ArrayView(C.Structure):
    """ah_array_view / orc_view (identical layout)."""
    _fields_ = [
        ("type", C.c_int32),
        ("length", C.c_int64),
        ("null_count", C.c_int64),
        ("values", C.c_void_p),
        ("values_bit_offset", C.c_int64),
        ("validity", C.c_void_p),
        ("validity_bit_offset", C.c_int64),
        ("offsets", C.c_void_p),
    ]


class ArrayOut(C.Structure):
    """ah_array_out / orc_out (identical layout)."""
    _fields_ = [
        ("type", C.c_int32),
        ("length", C.c_int64),
        ("null_count", C.c_int64),
        ("values", C.c_void_p),
        ("values_bytes", C.c_int64),
        ("values_bit_offset", C.c_int64),
        ("validity", C.c_void_p),
        ("validity_bytes", C.c_int64),
        ("validity_bit_offset", C.c_int64),
        ("offsets", C.c_void_p),
        ("offsets_bytes", C.c_int64),
        ("flags", C.c_int32),
    ]


class FilterTerm(C.Structure):
    """ah_filter_term"""
    _fields_ = [("op", C.c_int32), ("lhs", C.POINTER(ArrayView)), ("lhs_is_scalar", C.c_int32), ("rhs", C.POINTER(ArrayView)),
                ("rhs_is_scalar", C.c_int32)]


class Scalar(C.Structure):
    """ah_scalar / orc_scalar (identical layout)."""
    _fields_ = [("type", C.c_int32), ("is_valid", C.c_int32), ("bytes", C.c_uint8 * 32)]


class ContextStats(C.Structure):
    """ah_context_stats_t"""
    _fields_ = [(n, C.c_int64) for n in ("live_bytes", "high_water_bytes", "cached_bytes", "reserved_high_water_bytes",
                                         "allocated_bytes_total", "freed_bytes_total", "alloc_calls", "free_calls", "pool_hits",
                                         "device_malloc_calls", "host_to_device_bytes", "device_to_host_bytes")]


class ExchangeStats(C.Structure):
    """ah_exchange_stats"""
    _fields_ = [("peers", C.c_int32), ("bytes_to_each_peer", C.c_int64), ("bytes_received", C.c_int64),
                ("counts_ms", C.c_double), ("total_ms", C.c_double)]


class IpcField(C.Structure):
    """ah_ipc_field"""
    _fields_ = [("name", C.c_char_p), ("format", C.c_char_p), ("nullable", C.c_int32)]


class IpcBlock(C.Structure):
    """ah_ipc_block == File.fbs `struct Block`"""
    _fields_ = [("offset", C.c_int64), ("meta_data_length", C.c_int32), ("reserved_", C.c_int32),
                ("body_length", C.c_int64)]


class FFI_ArrowSchema(C.Structure):
    """struct ArrowSchema (arrow-schema/src/ffi.rs:76-98)."""


FFI_ArrowSchema._fields_ = [
    ("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_void_p), ("flags", C.c_int64),
    ("n_children", C.c_int64), ("children", C.POINTER(C.POINTER(FFI_ArrowSchema))),
    ("dictionary", C.POINTER(FFI_ArrowSchema)), ("release", C.c_void_p), ("private_data", C.c_void_p)]


class FFI_ArrowArray(C.Structure):
    """struct ArrowArray (arrow-data/src/ffi.rs:37-66)."""


FFI_ArrowArray._fields_ = [
    ("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
    ("n_children", C.c_int64), ("buffers", C.POINTER(C.c_void_p)),
    ("children", C.POINTER(C.POINTER(FFI_ArrowArray))), ("dictionary", C.POINTER(FFI_ArrowArray)),
    ("release", C.c_void_p), ("private_data", C.c_void_p)]

class FFI_ArrowDeviceArray(C.Structure):
    """struct ArrowDeviceArray (Arrow C Device Data Interface)."""
    _fields_ = [("array", FFI_ArrowArray), ("device_id", C.c_int64), ("device_type", C.c_int32), ("sync_event", C.c_void_p),
                ("reserved", C.c_int64 * 3)]


ARROW_DEVICE_ROCM = 10

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
FREE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t)

# every symbol include/arrow_hip.h declares: (restype, argtypes)
_P = C.c_void_p
_VIEW = C.POINTER(ArrayView)
_OUT = C.POINTER(ArrayOut)
SIGNATURES = {
    "ah_device_count": (C.c_int32, []),
    "ah_context_create": (C.c_int32, [C.c_int, C.POINTER(_P)]),
    "ah_context_destroy": (None, [_P]),
    "ah_context_set_allocator": (None, [_P, ALLOC_FN, FREE_FN, _P]),
    "ah_context_set_stream": (None, [_P, _P]),
    "ah_context_set_deferred": (None, [_P, C.c_int32]),
    "ah_context_deferred": (C.c_int32, [_P]),
    "ah_array_resolve": (C.c_int32, [_P, _P]),
    "ah_context_stream": (_P, [_P]),
    "ah_last_error": (C.c_char_p, [_P]),
    "ah_array_release": (None, [_P, _OUT]),
    "ah_version": (C.c_char_p, []),
    "ah_device_alloc": (C.c_int32, [_P, C.c_size_t, C.POINTER(_P)]),
    "ah_device_free": (None, [_P, _P]),
    "ah_memcpy_htod": (C.c_int32, [_P, _P, _P, C.c_size_t]),
    "ah_memcpy_dtoh": (C.c_int32, [_P, _P, _P, C.c_size_t]),
    "ah_memcpy_dtod": (C.c_int32, [_P, _P, _P, C.c_size_t]),
    "ah_memset": (C.c_int32, [_P, _P, C.c_int, C.c_size_t]),
    "ah_synchronize": (C.c_int32, [_P]),
    "ah_context_stats": (C.c_int32, [_P, C.POINTER(ContextStats), C.c_int32]),
    "ah_graph_begin": (C.c_int32, [_P]),
    "ah_graph_end": (C.c_int32, [_P, C.POINTER(_P)]),
    "ah_graph_node_count": (C.c_int32, [_P]),
    "ah_graph_launch": (C.c_int32, [_P, _P]),
    "ah_graph_destroy": (None, [_P, _P]),
    "ah_pool_trim": (None, [_P]),
    "ah_filter": (C.c_int32, [_P, _VIEW, _VIEW, _OUT]),
    "ah_filter_predicate_build": (C.c_int32, [_P, _VIEW, C.POINTER(_P)]),
    "ah_filter_predicate_build_expr": (C.c_int32, [_P, C.c_int32, C.POINTER(FilterTerm), C.POINTER(C.c_int32), C.POINTER(_P)]),
    "ah_filter_expr": (C.c_int32, [_P, C.c_int32, C.POINTER(FilterTerm), C.POINTER(C.c_int32), _VIEW, _OUT]),
    "ah_array_shrink_to_fit": (C.c_int32, [_P, _OUT]),
    "ah_filter_predicate_count": (C.c_int64, [_P]),
    "ah_filter_predicate_apply": (C.c_int32, [_P, _P, _VIEW, _OUT]),
    "ah_filter_predicate_free": (None, [_P, _P]),
    "ah_filter_record_batch": (C.c_int32, [_P, C.c_int32, _VIEW, _VIEW, _OUT, C.POINTER(C.c_int64)]),
    "ah_filter_predicate_apply_into": (C.c_int32, [_P, _P, _VIEW, _P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "ah_copy_rows_into": (C.c_int32, [_P, _VIEW, C.c_int64, C.c_int64, _P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "ah_filter_predicate_apply_into_acc": (C.c_int32, [_P, _P, _VIEW, _P, _P, C.c_int64, _P]),
    "ah_copy_rows_into_acc": (C.c_int32, [_P, _VIEW, C.c_int64, C.c_int64, _P, _P, C.c_int64, _P]),
    "ah_read_words": (C.c_int32, [_P, _P, C.c_int32, C.POINTER(C.c_uint64), C.c_int32]),
    "ah_coalescer_create": (C.c_int32, [_P, C.c_int32, C.POINTER(C.c_int32), C.c_int64, C.POINTER(_P)]),
    "ah_coalescer_destroy": (None, [_P, _P]),
    "ah_coalescer_set_biggest_coalesce_batch_size": (None, [_P, C.c_int64]),
    "ah_coalescer_buffered_rows": (C.c_int64, [_P]),
    "ah_coalescer_completed_count": (C.c_int32, [_P]),
    "ah_coalescer_push_batch": (C.c_int32, [_P, _P, _VIEW, C.c_int64, C.c_uint64, C.POINTER(C.c_int32)]),
    "ah_coalescer_push_batch_with_filter": (C.c_int32, [_P, _P, _VIEW, C.c_int64, _VIEW, C.c_uint64, C.POINTER(C.c_int32)]),
    "ah_coalescer_push_batches_with_filters": (C.c_int32, [_P, _P, C.c_int32, _VIEW, C.POINTER(C.c_int64), _VIEW,
                                                           C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    "ah_coalescer_push_batches_with_filters_begin": (C.c_int32, [_P, _P, C.c_int32, _VIEW, C.POINTER(C.c_int64), _VIEW,
                                                                C.POINTER(C.c_uint64), C.POINTER(_P)]),
    "ah_coalescer_push_batches_with_filters_end": (C.c_int32, [_P, _P, _P, C.POINTER(C.c_int32)]),
    "ah_coalescer_declare_view_buffers": (C.c_int32, [_P, _P, C.c_int32, C.POINTER(C.c_int32)]),
    "ah_coalescer_completed_batch_sources": (C.c_int32, [_P, _P, C.POINTER(C.c_uint64), C.c_int32, C.POINTER(C.c_int32)]),
    "ah_coalescer_push_batch_with_indices": (C.c_int32, [_P, _P, _VIEW, C.c_int64, _VIEW]),
    "ah_filter_predicates_build": (C.c_int32, [_P, C.c_int32, _VIEW, C.POINTER(_P)]),
    "ah_coalescer_finish_buffered_batch": (C.c_int32, [_P, _P]),
    "ah_coalescer_next_completed_batch": (C.c_int32, [_P, _P, _OUT, C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]),
    "ah_coalescer_next_completed_batches": (C.c_int32, [_P, _P, C.c_int32, _OUT, C.POINTER(C.c_int64), C.POINTER(C.c_uint64),
                                                        C.POINTER(C.c_int32)]),
    "ah_arrays_release": (None, [_P, _OUT, C.c_int64]),
    "ah_coalescer_push_abort": (None, [_P, _P, _P]),
    "ah_take": (C.c_int32, [_P, _VIEW, _VIEW, C.c_int32, _OUT]),
    "ah_arith_binary": (C.c_int32, [_P, C.c_int32, _VIEW, C.c_int32, _VIEW, C.c_int32, _OUT]),
    "ah_arith_with_types": (C.c_int32, [_P, C.c_int32, _VIEW, C.c_int32, C.POINTER(DataTypeDesc), _VIEW, C.c_int32,
                                        C.POINTER(DataTypeDesc), _OUT, C.POINTER(DataTypeDesc)]),
    "ah_bitwise_not": (C.c_int32, [_P, _VIEW, _OUT]),
    "ah_arith_neg": (C.c_int32, [_P, _VIEW, C.c_int32, _OUT]),
    "ah_compare": (C.c_int32, [_P, C.c_int32, _VIEW, C.c_int32, _VIEW, C.c_int32, _OUT]),
    "ah_compare_with_types": (C.c_int32, [_P, C.c_int32, _VIEW, C.c_int32, C.POINTER(DataTypeDesc), _VIEW, C.c_int32,
                                          C.POINTER(DataTypeDesc), _OUT]),
    "ah_boolean_binary": (C.c_int32, [_P, C.c_int32, _VIEW, _VIEW, _OUT]),
    "ah_boolean_unary": (C.c_int32, [_P, C.c_int32, _VIEW, _OUT]),
    "ah_nullif": (C.c_int32, [_P, _VIEW, _VIEW, _OUT]),
    "ah_cast": (C.c_int32, [_P, _VIEW, C.c_int32, C.c_int32, _OUT]),
    "ah_cast_chain": (C.c_int32, [_P, _VIEW, C.c_int32, C.POINTER(C.c_int32), C.c_int32, _OUT]),
    "ah_can_cast_types": (C.c_int32, [C.c_int32, C.c_int32]),
    "ah_cast_with_types": (C.c_int32, [_P, _VIEW, C.POINTER(DataTypeDesc), C.POINTER(DataTypeDesc), C.c_int32, _OUT]),
    "ah_can_cast_data_types": (C.c_int32, [C.POINTER(DataTypeDesc), C.POINTER(DataTypeDesc)]),
    "ah_concat": (C.c_int32, [_P, C.c_int32, _VIEW, _OUT]),
    "ah_bitmap_set_bits": (C.c_int32, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]),
    "ah_comm_unique_id": (C.c_int32, [_P, C.c_char_p]),
    "ah_comm_create": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_char_p, C.POINTER(_P)]),
    "ah_comm_destroy": (None, [_P, _P]),
    "ah_comm_rank": (C.c_int32, [_P]),
    "ah_comm_world": (C.c_int32, [_P]),
    "ah_comm_barrier": (C.c_int32, [_P, _P]),
    "ah_comm_allreduce_max_f64": (C.c_int32, [_P, _P, C.POINTER(C.c_double), C.c_int32]),
    "ah_all_gatherv": (C.c_int32, [_P, _P, _VIEW, _OUT, C.POINTER(ExchangeStats)]),
    "ah_all_gather_columns": (C.c_int32, [_P, _P, C.c_int32, _VIEW, _OUT, C.POINTER(ExchangeStats)]),
    "ah_all_gather_columns_begin": (C.c_int32, [_P, _P, C.c_int32, _VIEW, C.POINTER(_P)]),
    "ah_all_gather_columns_end": (C.c_int32, [_P, _P, _P, _OUT, C.POINTER(ExchangeStats)]),
    "ah_bitmap_concat": (C.c_int32, [_P, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int64), C.POINTER(C.c_int64), _P,
                                     C.POINTER(C.c_int64)]),
    "ah_count_set_bits": (C.c_int32, [_P, _P, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]),
    "ah_profile_enable": (None, [_P, C.c_int32]),
    "ah_profile_reset": (None, [_P]),
    "ah_profile_get": (C.c_int32, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ah_gen_uniform_i64": (C.c_int32, [_P, _P, C.c_int64, C.c_uint64, C.c_int64, C.c_int64, C.c_int64]),
    "ah_gen_uniform_i32": (C.c_int32, [_P, _P, C.c_int64, C.c_uint64, C.c_int64]),
    "ah_gen_uniform_f64": (C.c_int32, [_P, _P, C.c_int64, C.c_uint64, C.c_double, C.c_double, C.c_int64]),
    "ah_gen_uniform_u32": (C.c_int32, [_P, _P, C.c_int64, C.c_uint64, C.c_uint32, C.c_int64]),
    "ah_gen_uniform_f32": (C.c_int32, [_P, _P, C.c_int64, C.c_uint64, C.c_float, C.c_float, C.c_int64]),
    "ah_gen_uniform_small": (C.c_int32, [_P, _P, C.c_int32, C.c_int64, C.c_uint64, C.c_int64]),
    "ah_gen_iota_u32": (C.c_int32, [_P, _P, C.c_int64, C.c_uint32]),
    "ah_gen_bernoulli_bits": (C.c_int32, [_P, _P, C.c_int64, C.c_uint64, C.c_double, C.c_int64]),
    "ah_zero_null_slots": (C.c_int32, [_P, _P, C.c_int32, _P, C.c_int64]),
    "ah_aggregate": (C.c_int32, [_P, C.c_int32, _VIEW, C.POINTER(Scalar)]),
    "ah_interleave": (C.c_int32, [_P, C.c_int32, _VIEW, _VIEW, _VIEW, _OUT]),
    "ah_zip": (C.c_int32, [_P, _VIEW, _VIEW, C.c_int32, _VIEW, C.c_int32, _OUT]),
    "ah_sort_to_indices": (C.c_int32, [_P, _VIEW, C.c_int32, C.c_int32, C.c_int64, _OUT]),
    "ah_rank": (C.c_int32, [_P, _VIEW, C.c_int32, C.c_int32, _OUT]),
    "ah_shift": (C.c_int32, [_P, _VIEW, C.c_int64, _OUT]),
    "ah_lexsort_to_indices": (C.c_int32, [_P, C.c_int32, _VIEW, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int64, _OUT]),
    "ah_selection_and_then": (C.c_int32, [_P, _VIEW, _VIEW, _OUT]),
    "ah_selection_combine": (C.c_int32, [_P, C.c_int32, _VIEW, _VIEW, _OUT]),
    "ah_selection_boundaries": (C.c_int32, [_P, _VIEW, _OUT]),
    "ah_selection_from_boundaries": (C.c_int32, [_P, _VIEW, C.c_int64, _OUT]),
    "ah_selection_find_nth_set_bit": (C.c_int32, [_P, _VIEW, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]),
    "ah_ipc_schema_message": (C.c_int32, [_P, C.c_int32, C.POINTER(IpcField), C.c_int32, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "ah_ipc_decode_schema": (C.c_int32, [_P, C.c_char_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.POINTER(IpcField))]),
    "ah_ipc_encode_batch": (C.c_int32, [_P, C.c_int32, _VIEW, C.c_int64, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int64),
                                        C.POINTER(_P), C.POINTER(C.c_int64)]),
    "ah_ipc_decode_batch": (C.c_int32, [_P, C.c_char_p, C.c_int64, _P, C.c_int64, C.c_int32, C.POINTER(IpcField), _OUT,
                                        C.POINTER(C.c_int64)]),
    "ah_ipc_message_info": (C.c_int32, [_P, C.c_char_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "ah_string_like": (C.c_int32, [_P, C.c_int32, C.POINTER(ArrayView), C.POINTER(ArrayView), C.c_int32, C.POINTER(ArrayOut)]),
    "ah_string_length": (C.c_int32, [_P, C.POINTER(ArrayView), C.c_int32, C.POINTER(ArrayOut)]),
    "ah_ipc_file_footer": (C.c_int32, [_P, C.c_int32, C.POINTER(IpcField), C.c_int32, C.POINTER(IpcBlock),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "ah_ipc_decode_footer": (C.c_int32, [_P, C.c_char_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                         C.POINTER(C.POINTER(IpcField)), C.POINTER(C.c_int32),
                                         C.POINTER(C.POINTER(IpcBlock))]),
    "ah_host_free": (None, [_P]),
    "ah_type_from_format": (C.c_int32, [_P, C.c_char_p, C.POINTER(C.c_int32)]),
    "ah_format_of_type": (C.c_char_p, [C.c_int32]),
    "ah_import_c_data": (C.c_int32, [_P, C.POINTER(FFI_ArrowArray), C.POINTER(FFI_ArrowSchema), _OUT]),
    "ah_export_c_device_data": (C.c_int32, [_P, _VIEW, _OUT, C.c_char_p, C.POINTER(FFI_ArrowDeviceArray), C.POINTER(FFI_ArrowSchema)]),
    "ah_import_c_device_data": (C.c_int32, [_P, C.POINTER(FFI_ArrowDeviceArray), C.POINTER(FFI_ArrowSchema), _VIEW]),
    "ah_export_c_data": (C.c_int32, [_P, _VIEW, C.c_char_p, C.POINTER(FFI_ArrowArray), C.POINTER(FFI_ArrowSchema)]),
}

_lib = None


def load():
    """Load libarrow_hip.so, binding every declared symbol. Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C arrow-rs_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
