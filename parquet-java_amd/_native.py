"""ctypes binding of libpqdecode_hip.so (the C ABI declared in include/pqd.h).

The library is the product: if it is missing this module raises at import, there is no CPU
fallback.  torch is imported first so that the process holds a single HIP runtime (torch ships
its own libamdhip64 with the same soname).
"""
import ctypes as C
import os

try:  # plumbing only: device memory, streams, torch.distributed
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PQD_LIB_PATH") or os.path.join(_HERE, "lib", "libpqdecode_hip.so")  # override: development builds

OK, EOF, BAD_ARG, PAST_END, DECODE, DICT_OOB, HIP_ERROR, NO_DEVICE = range(8)
ENC_RLE, ENC_RLE_DICTIONARY, ENC_DELTA_BINARY_PACKED, ENC_BYTE_STREAM_SPLIT, ENC_BIT_PACKED = 1, 2, 3, 4, 5
ENC_PLAIN, ENC_PLAIN_BOOLEAN = 6, 7
ENC_DELTA_LENGTH_BYTE_ARRAY, ENC_DELTA_BYTE_ARRAY, ENC_PLAIN_BYTE_ARRAY = 8, 9, 10
F_LENGTH_PREFIX, F_WIDTH_PREFIX = 1, 2
TYPE_BOOLEAN, TYPE_INT32, TYPE_INT64, TYPE_INT96, TYPE_FLOAT, TYPE_DOUBLE, TYPE_BYTE_ARRAY, TYPE_FIXED_LEN_BYTE_ARRAY = 0, 1, 2, 3, 4, 5, 6, 7
PAGE_DATA, PAGE_INDEX, PAGE_DICTIONARY, PAGE_DATA_V2 = 0, 1, 2, 3
VIEW_INVALID = 0xFFFFFFFFFFFFFFFF
CODEC_UNCOMPRESSED, CODEC_SNAPPY, CODEC_GZIP, CODEC_LZO, CODEC_BROTLI, CODEC_LZ4, CODEC_ZSTD, CODEC_LZ4_RAW = range(8)


class PageDesc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("out", C.c_void_p), ("dict", C.c_void_p),
                ("data_len", C.c_uint32), ("num_values", C.c_uint32), ("dict_len", C.c_uint32),
                ("encoding", C.c_uint8), ("elem_size", C.c_uint8), ("bit_width", C.c_uint8), ("flags", C.c_uint8)]


class PageResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("values_written", C.c_uint32), ("bytes_consumed", C.c_uint32),
                ("reserved", C.c_uint32)]


class DictInfo(C.Structure):
    _fields_ = [("entries", C.c_void_p), ("bytes", C.c_void_p), ("n_entries", C.c_uint32),
                ("elem_size", C.c_uint32), ("bytes_len", C.c_uint32), ("physical_type", C.c_uint32)]


class BinaryPageDesc(C.Structure):
    _fields_ = [("views", C.c_void_p), ("dict_bytes", C.c_void_p), ("offsets", C.c_void_p), ("bytes", C.c_void_p),
                ("num_values", C.c_uint32), ("dict_bytes_len", C.c_uint32), ("bytes_capacity", C.c_uint32),
                ("reserved", C.c_uint32)]


class ValidityDesc(C.Structure):
    _fields_ = [("def_levels", C.c_void_p), ("validity", C.c_void_p), ("value_index", C.c_void_p),
                ("n_levels", C.c_uint32), ("max_def", C.c_int32)]


class ScatterDesc(C.Structure):
    _fields_ = [("values", C.c_void_p), ("validity", C.c_void_p), ("out", C.c_void_p),
                ("n_slots", C.c_uint32), ("n_values", C.c_uint32)]


class OffsetsDesc(C.Structure):
    _fields_ = [("rep_levels", C.c_void_p), ("def_levels", C.c_void_p), ("offsets", C.c_void_p), ("row_valid", C.c_void_p),
                ("n_levels", C.c_uint32), ("rows_capacity", C.c_uint32), ("slot_def", C.c_int32), ("row_def", C.c_int32)]


class DeltaBytesDesc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("offsets", C.c_void_p), ("bytes", C.c_void_p), ("work", C.c_void_p),
                ("prev_offsets", C.c_void_p), ("prev_bytes", C.c_void_p),
                ("data_len", C.c_uint32), ("num_values", C.c_uint32), ("bytes_capacity", C.c_uint32),
                ("encoding", C.c_uint32), ("prev_count", C.c_uint32), ("reserved", C.c_uint32)]


def delta_bytes_work(num_values):
    """PQD_DELTA_BYTES_WORK: bytes of scratch a delta byte array page needs."""
    return 64 + 8 * num_values


class DecompressDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("src_len", C.c_uint32), ("dst_len", C.c_uint32),
                ("codec", C.c_uint32), ("reserved", C.c_uint32)]


class ChunkPage(C.Structure):
    _fields_ = [("header_offset", C.c_uint32), ("body_offset", C.c_uint32), ("compressed_size", C.c_uint32),
                ("uncompressed_size", C.c_uint32), ("num_values", C.c_uint32), ("num_nulls", C.c_uint32),
                ("num_rows", C.c_uint32), ("rep_levels_len", C.c_uint32), ("def_levels_len", C.c_uint32), ("crc", C.c_uint32),
                ("type", C.c_uint8), ("encoding", C.c_uint8), ("def_encoding", C.c_uint8), ("rep_encoding", C.c_uint8),
                ("is_compressed", C.c_uint8), ("has_crc", C.c_uint8), ("reserved", C.c_uint8 * 2)]


class ColumnMeta(C.Structure):
    _fields_ = [("physical_type", C.c_int32), ("type_length", C.c_int32), ("max_def", C.c_int32), ("max_rep", C.c_int32)]


class ChunkOutputs(C.Structure):
    _fields_ = [("bodies", C.c_void_p), ("rep_out", C.c_void_p), ("def_out", C.c_void_p), ("val_out", C.c_void_p),
                ("level_base", C.c_void_p), ("value_base", C.c_void_p), ("dict", C.c_void_p), ("dict_len", C.c_uint32),
                ("reserved", C.c_uint32)]


assert C.sizeof(PageDesc) == 40 and C.sizeof(PageResult) == 16 and C.sizeof(BinaryPageDesc) == 48
assert C.sizeof(ChunkPage) == 48 and C.sizeof(ColumnMeta) == 16 and C.sizeof(ChunkOutputs) == 64
assert C.sizeof(ValidityDesc) == 32 and C.sizeof(ScatterDesc) == 32 and C.sizeof(OffsetsDesc) == 48

EXPORTS = {
    "pqd_version": ([], C.c_int),
    "pqd_status_string": ([C.c_int], C.c_char_p),
    "pqd_last_error": ([], C.c_char_p),
    "pqd_device_count": ([], C.c_int),
    "pqd_init": ([C.c_int], C.c_int),
    "pqd_dev_alloc": ([C.POINTER(C.c_void_p), C.c_size_t], C.c_int),
    "pqd_dev_free": ([C.c_void_p], C.c_int),
    "pqd_host_alloc_pinned": ([C.POINTER(C.c_void_p), C.c_size_t], C.c_int),
    "pqd_host_free_pinned": ([C.c_void_p], C.c_int),
    "pqd_alloc_pinned": ([C.POINTER(C.c_void_p), C.c_size_t], C.c_int),
    "pqd_free_pinned": ([C.c_void_p], C.c_int),
    "pqd_memcpy_h2d": ([C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p], C.c_int),
    "pqd_memcpy_d2h": ([C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p], C.c_int),
    "pqd_memset_dev": ([C.c_void_p, C.c_int, C.c_size_t, C.c_void_p], C.c_int),
    "pqd_stream_create": ([C.POINTER(C.c_void_p)], C.c_int),
    "pqd_stream_destroy": ([C.c_void_p], C.c_int),
    "pqd_stream_sync": ([C.c_void_p], C.c_int),
    "pqd_event_create": ([C.POINTER(C.c_void_p)], C.c_int),
    "pqd_event_destroy": ([C.c_void_p], C.c_int),
    "pqd_event_record": ([C.c_void_p, C.c_void_p], C.c_int),
    "pqd_event_sync": ([C.c_void_p], C.c_int),
    "pqd_set_phase_timing": ([C.c_int], C.c_int),
    "pqd_set_phase_series": ([C.c_int], C.c_int),
    "pqd_phase_ms_mean": ([C.POINTER(C.c_float), C.POINTER(C.c_int)], C.c_int),
    "pqd_event_elapsed_ms": ([C.c_void_p, C.c_void_p, C.POINTER(C.c_float)], C.c_int),
    "pqd_unpack_i32": ([C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t], C.c_int),
    "pqd_unpack_i32_dev": ([C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p], C.c_int),
    "pqd_unpack_i64": ([C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t], C.c_int),
    "pqd_unpack_i64_dev": ([C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p], C.c_int),
    "pqd_decode_pages_dev": ([C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "pqd_decode_pages_dev_noalloc": ([C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "pqd_decode_pages": ([C.POINTER(PageDesc), C.POINTER(PageResult), C.c_uint32, C.c_void_p], C.c_int),
    "pqd_decode_page": ([C.POINTER(PageDesc), C.POINTER(PageResult)], C.c_int),
    "pqd_decode_page_flat": ([C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                              C.c_uint32, C.c_void_p, C.POINTER(PageResult)], C.c_int),
    "pqd_thread_release": ([], C.c_int),
    "pqd_binary_dict_views": ([C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_void_p], C.c_int),
    "pqd_dict_create": ([C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p)], C.c_int),
    "pqd_dict_get": ([C.c_void_p, C.POINTER(DictInfo)], C.c_int),
    "pqd_dict_destroy": ([C.c_void_p], C.c_int),
    "pqd_binary_materialize_dev": ([C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p], C.c_int),
    "pqd_decode_binary_pages_dev": ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p], C.c_int),
    "pqd_def_levels_to_validity_dev": ([C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p], C.c_int),
    "pqd_scatter_nulls_dev": ([C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p], C.c_int),
    "pqd_rep_levels_to_offsets_dev": ([C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p], C.c_int),
    "pqd_chunk_walk": ([C.c_void_p, C.c_size_t, C.c_int64, C.c_int, C.POINTER(ChunkPage), C.c_uint32, C.POINTER(C.c_uint32)], C.c_int),
    "pqd_crc32": ([C.c_uint32, C.c_void_p, C.c_size_t], C.c_uint32),
    "pqd_chunk_sections_dev": ([C.c_void_p, C.c_uint32, C.POINTER(ColumnMeta), C.POINTER(ChunkOutputs), C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    "pqd_chunk_set_value_counts_dev": ([C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p], C.c_int),
    "pqd_delta_byte_array_dev": ([C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p], C.c_int),
    "pqd_decode_delta_byte_array_page": ([C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                          C.c_void_p, C.c_uint32, C.POINTER(PageResult), C.POINTER(C.c_uint64)], C.c_int),
    "pqd_decompress_pages_dev": ([C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p], C.c_int),
    "pqd_codec_supported": ([C.c_int], C.c_int),
    "pqd_chunk_decompress_dev": ([C.c_void_p, C.POINTER(ChunkPage), C.c_uint32, C.c_int, C.c_void_p, C.c_size_t,
                                  C.POINTER(ChunkPage), C.c_void_p, C.c_void_p, C.c_void_p], C.c_int),
    "pqd_decompress_page": ([C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(PageResult)], C.c_int),
    "pqd_decode_binary_page": ([C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                C.POINTER(PageResult)], C.c_int),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950). "
        "There is no CPU fallback for the decode path.")

lib = C.CDLL(LIB_PATH)
for _name, (_args, _res) in EXPORTS.items():
    _fn = getattr(lib, _name)  # AttributeError if the library does not export a declared symbol
    _fn.argtypes = _args
    _fn.restype = _res


class PqdError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        msg = lib.pqd_status_string(status).decode()
        detail = lib.pqd_last_error().decode()
        super().__init__(f"{where}: {msg}" + (f" [{detail}]" if detail else ""))


def check(status, where=""):
    if status != OK:
        raise PqdError(status, where)
