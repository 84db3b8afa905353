"""ctypes front-end of the CPU oracle (``libpqd_oracle.so``).

TEST INFRASTRUCTURE ONLY: the oracle is the checker the HIP kernels are compared
with.  Only tests, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` import this module; the product package never does.

All functions take/return numpy arrays and return ``(status, ...)`` tuples with the
status codes of ``pqd_oracle.h`` (numerically equal to the ``PQD_*`` codes).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpqd_oracle.so")

OK, EOF, BAD_ARG, PAST_END, DECODE, DICT_OOB = 0, 1, 2, 3, 4, 5


def build(force=False):
    """Compile the oracle with gcc (a few hundred ms)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libpqd_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_szp = C.POINTER(C.c_size_t)


def _declare(L):
    L.orc_unpack8_32.argtypes = [C.c_int, C.c_int, _u8p, _i32p]
    L.orc_unpack8_32.restype = None
    L.orc_unpack8_64.argtypes = [C.c_int, C.c_int, _u8p, _i64p]
    L.orc_unpack8_64.restype = None
    L.orc_pack8_32.argtypes = [C.c_int, C.c_int, _i32p, _u8p]
    L.orc_pack8_32.restype = None
    L.orc_pack8_64.argtypes = [C.c_int, C.c_int, _i64p, _u8p]
    L.orc_pack8_64.restype = None
    L.orc_read_batch_i32.argtypes = [C.c_int, _u8p, C.c_size_t, _i32p, C.c_size_t, _szp]
    L.orc_read_batch_i64.argtypes = [C.c_int, _u8p, C.c_size_t, _i64p, C.c_size_t, _szp]
    L.orc_rle_decode.argtypes = [C.c_int, _u8p, C.c_size_t, _i32p, C.c_size_t, _szp, _szp]
    L.orc_rle_prefixed_decode.argtypes = [C.c_int, _u8p, C.c_size_t, _i32p, C.c_size_t, _szp, _szp]
    L.orc_rle_encode.argtypes = [C.c_int, _i32p, C.c_size_t, _u8p, C.c_size_t]
    L.orc_rle_encode.restype = C.c_long
    L.orc_dict_decode_ids.argtypes = [_u8p, C.c_size_t, _i32p, C.c_size_t, _szp]
    L.orc_dict_decode.argtypes = [_u8p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                                  C.c_size_t, _szp]
    L.orc_dict_encode_ids.argtypes = [_i32p, C.c_size_t, C.c_size_t, _u8p, C.c_size_t]
    L.orc_dict_encode_ids.restype = C.c_long
    L.orc_delta_decode.argtypes = [_u8p, C.c_size_t, _i64p, C.c_size_t, _szp, _szp]
    L.orc_delta_encode_i64.argtypes = [_i64p, C.c_size_t, C.c_int, C.c_int, _u8p, C.c_size_t]
    L.orc_delta_encode_i64.restype = C.c_long
    L.orc_delta_encode_i32.argtypes = [_i32p, C.c_size_t, C.c_int, C.c_int, _u8p, C.c_size_t]
    L.orc_delta_encode_i32.restype = C.c_long
    L.orc_binary_dict_views.argtypes = [_u8p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
    L.orc_binary_materialize.argtypes = [C.c_void_p, C.c_size_t, _u8p, C.c_size_t, _i32p, _u8p, C.c_size_t, _szp, _szp]
    L.orc_bss_decode.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_size_t, _u8p, _szp]
    L.orc_bss_encode.argtypes = [_u8p, C.c_size_t, C.c_int, _u8p]
    L.orc_bss_encode.restype = None
    L.orc_bitpacked_levels_decode.argtypes = [C.c_int, _u8p, C.c_size_t, _i32p, C.c_size_t, _szp]
    L.orc_plain_decode.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_size_t, _u8p, _szp]
    L.orc_plain_boolean_decode.argtypes = [_u8p, C.c_size_t, C.c_size_t, _u8p, _szp, _szp]
    L.orc_def_validity.argtypes = [_i32p, C.c_size_t, C.c_int32, _u8p, _i32p, _szp]
    L.orc_def_validity.restype = None
    L.orc_scatter_nulls.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, _u8p, _szp, _szp]
    L.orc_rep_offsets.argtypes = [_i32p, _i32p, C.c_size_t, C.c_int32, C.c_int32, C.c_size_t, _i32p, _u8p, _szp, _szp]
    L.orc_snappy_decompress.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, _szp, _szp]
    L.orc_lz4_raw_decompress.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, _szp, _szp]
    L.orc_zstd_decompress.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, _szp, _szp]
    L.orc_gzip_decompress.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, _szp, _szp]
    L.orc_xxh64.argtypes = [_u8p, C.c_size_t]
    L.orc_xxh64.restype = C.c_uint64
    L.orc_delta_length_byte_array.argtypes = [_u8p, C.c_size_t, C.c_size_t, _i32p, _u8p, C.c_size_t, _szp, _szp, _szp]
    L.orc_plain_byte_array.argtypes = [_u8p, C.c_size_t, C.c_size_t, _i32p, _u8p, C.c_size_t, _szp, _szp, _szp]
    L.orc_delta_byte_array.argtypes = [_u8p, C.c_size_t, C.c_size_t, _u8p, C.c_size_t, _i32p, _u8p, C.c_size_t, _szp, _szp,
                                       _szp]


def _u8(a):
    a = np.ascontiguousarray(np.frombuffer(bytes(a), dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a,
                             dtype=np.uint8)
    if a.size == 0:
        a = np.zeros(1, np.uint8)[:0]
    return a


def _p(a, t):
    if a.size == 0:  # a valid non-null pointer for empty arrays
        return C.cast(C.create_string_buffer(8), t)
    return a.ctypes.data_as(t)


# ---------------------------------------------------------------- bit packing
def unpack8(w, data, msb_first=False, long=False):
    data = _u8(data)
    out = np.zeros(8, np.int64 if long else np.int32)
    f = lib().orc_unpack8_64 if long else lib().orc_unpack8_32
    f(int(msb_first), w, _p(data, _u8p), _p(out, _i64p if long else _i32p))
    return out


def pack8(w, values, msb_first=False, long=False):
    v = np.ascontiguousarray(values, np.int64 if long else np.int32)
    out = np.zeros(max(w, 1), np.uint8)
    f = lib().orc_pack8_64 if long else lib().orc_pack8_32
    f(int(msb_first), w, _p(v, _i64p if long else _i32p), _p(out, _u8p))
    return out[:w]


def pack(w, values, msb_first=False, long=False):
    """Pack a multiple-of-8 array with repeated pack8Values."""
    v = np.ascontiguousarray(values, np.int64 if long else np.int32)
    assert v.size % 8 == 0
    out = np.zeros(v.size // 8 * w, np.uint8)
    for g in range(v.size // 8):
        out[g * w:(g + 1) * w] = pack8(w, v[g * 8:(g + 1) * 8], msb_first, long)
    return out


def read_batch(w, data, n, long=False):
    data = _u8(data)
    out = np.zeros(n, np.int64 if long else np.int32)
    used = C.c_size_t(0)
    f = lib().orc_read_batch_i64 if long else lib().orc_read_batch_i32
    st = f(w, _p(data, _u8p), data.size, _p(out, _i64p if long else _i32p), n, C.byref(used))
    return st, out, used.value


# ---------------------------------------------------------------- RLE hybrid
def rle_decode(w, data, n, prefixed=False):
    data = _u8(data)
    out = np.zeros(n, np.int32)
    wr, used = C.c_size_t(0), C.c_size_t(0)
    f = lib().orc_rle_prefixed_decode if prefixed else lib().orc_rle_decode
    st = f(w, _p(data, _u8p), data.size, _p(out, _i32p), n, C.byref(wr), C.byref(used))
    return st, out[:wr.value], used.value


def rle_encode(w, values):
    v = np.ascontiguousarray(values, np.int32)
    cap = v.size * 5 + v.size // 8 + 64
    out = np.zeros(cap, np.uint8)
    n = lib().orc_rle_encode(w, _p(v, _i32p), v.size, _p(out, _u8p), cap)
    assert n >= 0
    return out[:n].copy()


# ---------------------------------------------------------------- dictionary
def dict_encode_ids(ids, dict_n):
    v = np.ascontiguousarray(ids, np.int32)
    cap = v.size * 5 + v.size // 8 + 64
    out = np.zeros(cap, np.uint8)
    n = lib().orc_dict_encode_ids(_p(v, _i32p), v.size, dict_n, _p(out, _u8p), cap)
    assert n >= 0
    return out[:n].copy()


def dict_decode_ids(page, n):
    page = _u8(page)
    out = np.zeros(n, np.int32)
    wr = C.c_size_t(0)
    st = lib().orc_dict_decode_ids(_p(page, _u8p), page.size, _p(out, _i32p), n, C.byref(wr))
    return st, out[:wr.value]


def dict_decode(page, dictionary, n):
    page = _u8(page)
    d = np.ascontiguousarray(dictionary)
    out = np.zeros(n, d.dtype)
    wr = C.c_size_t(0)
    st = lib().orc_dict_decode(_p(page, _u8p), page.size, d.ctypes.data_as(C.c_void_p) if d.size else None,
                               d.size, d.dtype.itemsize, out.ctypes.data_as(C.c_void_p) if n else None,
                               n, C.byref(wr))
    return st, out[:wr.value]


# ---------------------------------------------------------------- BINARY dictionaries
VIEW_INVALID = 0xFFFFFFFFFFFFFFFF


def binary_dict_page(values):
    """PLAIN-encoded BYTE_ARRAY dictionary page body: [len LE32][bytes] per entry
    (PlainValuesWriter.writeBytes as used by PlainBinaryDictionaryValuesWriter.toDictPageAndClose)."""
    out = bytearray()
    for v in values:
        out += len(v).to_bytes(4, "little") + bytes(v)
    return np.frombuffer(bytes(out), np.uint8).copy()


def binary_dict_views(page, n, fixed_length=0):
    page = _u8(page)
    views = np.zeros(n, np.uint64)
    st = lib().orc_binary_dict_views(_p(page, _u8p), page.size, n, fixed_length,
                                     views.ctypes.data_as(C.c_void_p))
    return st, views


def binary_materialize(views, dict_bytes, cap):
    views = np.ascontiguousarray(views, np.uint64)
    dict_bytes = _u8(dict_bytes)
    offsets = np.zeros(views.size + 1, np.int32)
    out = np.zeros(max(cap, 1), np.uint8)
    wr, bw = C.c_size_t(0), C.c_size_t(0)
    st = lib().orc_binary_materialize(views.ctypes.data_as(C.c_void_p), views.size, _p(dict_bytes, _u8p),
                                      dict_bytes.size, _p(offsets, _i32p), _p(out, _u8p), cap,
                                      C.byref(wr), C.byref(bw))
    return st, offsets[:wr.value + 1], out[:bw.value]


# ---------------------------------------------------------------- delta
def delta_encode(values, block=128, mini=4):
    v = np.ascontiguousarray(values)
    assert v.dtype in (np.int32, np.int64)
    cap = v.size * 10 + 4096
    out = np.zeros(cap, np.uint8)
    if v.dtype == np.int64:
        n = lib().orc_delta_encode_i64(_p(v, _i64p), v.size, block, mini, _p(out, _u8p), cap)
    else:
        n = lib().orc_delta_encode_i32(_p(v, _i32p), v.size, block, mini, _p(out, _u8p), cap)
    assert n >= 0
    return out[:n].copy()


def delta_decode(data, cap):
    data = _u8(data)
    out = np.zeros(max(cap, 1), np.int64)
    total, used = C.c_size_t(0), C.c_size_t(0)
    st = lib().orc_delta_decode(_p(data, _u8p), data.size, _p(out, _i64p), cap, C.byref(total), C.byref(used))
    n = total.value if st == OK else 0
    return st, out[:n], total.value, used.value


# ---------------------------------------------------------------- byte stream split
def bss_decode(data, k, bound):
    data = _u8(data)
    out = np.zeros(max(data.size, 1), np.uint8)
    n = C.c_size_t(0)
    st = lib().orc_bss_decode(_p(data, _u8p), data.size, k, bound, _p(out, _u8p), C.byref(n))
    return st, out[:n.value * k]


def bss_encode(raw, k):
    raw = _u8(raw)
    assert raw.size % k == 0
    out = np.zeros(max(raw.size, 1), np.uint8)
    lib().orc_bss_encode(_p(raw, _u8p), raw.size // k, k, _p(out, _u8p))
    return out[:raw.size]


# ---------------------------------------------------------------- legacy levels
def bitpacked_levels_decode(w, data, n):
    data = _u8(data)
    out = np.zeros(n, np.int32)
    used = C.c_size_t(0)
    st = lib().orc_bitpacked_levels_decode(w, _p(data, _u8p), data.size, _p(out, _i32p), n, C.byref(used))
    return st, out, used.value


# ---------------------------------------------------------------- PLAIN pages
def plain_decode(data, elem, n):
    data = _u8(data)
    out = np.zeros(max(n * elem, 1), np.uint8)
    wr = C.c_size_t(0)
    st = lib().orc_plain_decode(_p(data, _u8p), data.size, elem, n, _p(out, _u8p), C.byref(wr))
    return st, out[:wr.value * elem]


def plain_boolean_decode(data, n):
    data = _u8(data)
    out = np.zeros(max(n, 1), np.uint8)
    wr, used = C.c_size_t(0), C.c_size_t(0)
    st = lib().orc_plain_boolean_decode(_p(data, _u8p), data.size, n, _p(out, _u8p), C.byref(wr), C.byref(used))
    return st, out[:wr.value], used.value


# ---------------------------------------------------------------- levels -> columnar structure
def def_validity(def_levels, max_def, n=None):
    """(validity bitmap bytes, value_index int32[n], non-null count); def_levels None = required column."""
    if def_levels is None:
        lv, ptr = None, None
    else:
        lv = np.ascontiguousarray(def_levels, np.int32)
        n = lv.size
        ptr = _p(lv, _i32p)
    validity = np.zeros(max((n + 7) // 8, 1), np.uint8)
    index = np.zeros(max(n, 1), np.int32)
    cnt = C.c_size_t(0)
    lib().orc_def_validity(ptr, n, max_def, _p(validity, _u8p), _p(index, _i32p), C.byref(cnt))
    return validity[:(n + 7) // 8], index[:n], cnt.value


def scatter_nulls(values, validity, n_slots, elem):
    values = _u8(np.ascontiguousarray(values).view(np.uint8))
    out = np.zeros(max(n_slots * elem, 1), np.uint8)
    wr, bits = C.c_size_t(0), C.c_size_t(0)
    vptr = None if validity is None else _p(_u8(validity), _u8p)
    st = lib().orc_scatter_nulls(_p(values, _u8p), values.size // elem, vptr, n_slots, elem, _p(out, _u8p),
                                 C.byref(wr), C.byref(bits))
    return st, out[:n_slots * elem], wr.value, bits.value


def rep_offsets(rep, deff, slot_def, row_def, rows_cap):
    rep = np.ascontiguousarray(rep, np.int32)
    deff = np.ascontiguousarray(deff, np.int32)
    offsets = np.zeros(rows_cap + 1, np.int32)
    row_valid = np.zeros(max(rows_cap, 1), np.uint8)
    rows, slots = C.c_size_t(0), C.c_size_t(0)
    st = lib().orc_rep_offsets(_p(rep, _i32p), _p(deff, _i32p), rep.size, slot_def, row_def, rows_cap, _p(offsets, _i32p),
                               _p(row_valid, _u8p), C.byref(rows), C.byref(slots))
    k = min(rows.value, rows_cap)
    return st, offsets[:k + 1], row_valid[:k], rows.value, slots.value


# ---------------------------------------------------------------- PLAIN BYTE_ARRAY
def plain_byte_array(data, n, cap=None, in_place=False):
    """(status, offsets[written + 1], bytes, consumed) of a PLAIN page of BYTE_ARRAY values; in_place: the bytes
    are not copied (bytes comes back empty, the last offset still counts them)."""
    data = _u8(data)
    cap = data.size if cap is None else cap
    offsets = np.zeros(n + 1, np.int32)
    out = np.zeros(max(cap, 1), np.uint8)
    wr, used, nb = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    st = lib().orc_plain_byte_array(_p(data, _u8p), data.size, n, _p(offsets, _i32p), None if in_place else _p(out, _u8p), cap,
                                    C.byref(wr), C.byref(used), C.byref(nb))
    return st, offsets[:wr.value + 1], out[:0 if in_place else nb.value], used.value


# ---------------------------------------------------------------- DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY
def delta_length_byte_array(data, n, cap=None):
    """(status, offsets[written + 1], bytes, consumed) of a DELTA_LENGTH_BYTE_ARRAY values section."""
    data = _u8(data)
    cap = data.size if cap is None else cap
    offsets = np.zeros(n + 1, np.int32)
    out = np.zeros(max(cap, 1), np.uint8)
    wr, used, nb = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    st = lib().orc_delta_length_byte_array(_p(data, _u8p), data.size, n, _p(offsets, _i32p), _p(out, _u8p), cap, C.byref(wr),
                                           C.byref(used), C.byref(nb))
    return st, offsets[:wr.value + 1], out[:nb.value], used.value


def delta_byte_array(data, n, cap, prev=b""):
    """(status, offsets[written + 1], bytes, consumed) of a DELTA_BYTE_ARRAY values section."""
    data = _u8(data)
    prev = _u8(prev) if len(prev) else np.zeros(1, np.uint8)[:0]
    offsets = np.zeros(n + 1, np.int32)
    out = np.zeros(max(cap, 1), np.uint8)
    wr, used, nb = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    pv = np.ascontiguousarray(np.concatenate([prev, np.zeros(1, np.uint8)]))
    st = lib().orc_delta_byte_array(_p(data, _u8p), data.size, n, _p(pv, _u8p), prev.size, _p(offsets, _i32p), _p(out, _u8p),
                                    cap, C.byref(wr), C.byref(used), C.byref(nb))
    return st, offsets[:wr.value + 1], out[:nb.value], used.value


# ---------------------------------------------------------------- page decompression
def decompress(codec, data, size):
    """(status, bytes produced, consumed) of a SNAPPY / LZ4_RAW / ZSTD / GZIP page body that should hold `size` bytes."""
    data = _u8(data)
    out = np.zeros(max(size, 1), np.uint8)
    made, used = C.c_size_t(0), C.c_size_t(0)
    f = {"SNAPPY": lib().orc_snappy_decompress, "LZ4_RAW": lib().orc_lz4_raw_decompress, "ZSTD": lib().orc_zstd_decompress,
         "GZIP": lib().orc_gzip_decompress}[codec]
    st = f(_p(data, _u8p), data.size, _p(out, _u8p), size, C.byref(made), C.byref(used))
    return st, out[:made.value], used.value
