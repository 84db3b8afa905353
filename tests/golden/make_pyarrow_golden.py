#!/usr/bin/env python3
"""Writes small parquet files with pyarrow, cuts every column chunk out of them and stores the chunk
bytes with the values pyarrow itself reads back, in ``pyarrow_chunks.json``.  Run in the build
container (pyarrow is importable there; it never travels to the GPU box):

    python tests/golden/make_pyarrow_golden.py

pyarrow (Arrow C++) is NOT the reference: it is an independent implementation of the same format.
These fixtures exist because the reference's tree holds no byte vector for DELTA_BINARY_PACKED,
BYTE_STREAM_SPLIT, DataPage V2, DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY or compressed pages, and no
JVM is available to write some: with them the oracle's decoders are checked against pages that the
oracle's own encoders did not write.  The expected values are pyarrow's reading of its own file.

A fixture is one column chunk: the bytes from its first page header to its last page, the column's
physical type, maximum levels and codec, and the expected values (non-null values in order, plus the
null mask where the column is optional).
"""
import base64
import json
import os
import sys

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

HERE = os.path.dirname(os.path.abspath(__file__))
TMP = "/tmp/pqd_pyarrow_golden"


def specials32():
    return np.array([0x7FC00000, 0x7FC00001, 0xFFC00000, 0x7F800001, 0x80000000, 0, 0x00000001, 0x7F800000, 0xFF800000],
                    np.uint32).view(np.float32)


def specials64():
    return np.array([0x7FF8000000000000, 0x7FF8000000000001, 0xFFF8000000000000, 0x7FF0000000000001, 0x8000000000000000, 0,
                     1, 0x7FF0000000000000, 0xFFF0000000000000], np.uint64).view(np.float64)


def with_nulls(values, rng, frac, pa_type):
    mask = rng.random(len(values)) < frac
    return pa.array(values, type=pa_type, mask=mask)


def tables():
    rng = np.random.default_rng(20261111)
    n = 2000
    ts = (1_600_000_000_000_000_000 + np.cumsum(1000 + rng.integers(0, 1 << 12, n))).astype(np.int64)
    ts_jumps = ts.copy()
    ts_jumps[::97] -= 1 << 40  # negative deltas, wide miniblocks next to narrow ones
    wrap = np.where(np.arange(n) % 2 == 0, np.iinfo(np.int64).min, np.iinfo(np.int64).max).astype(np.int64)[:300]
    i32 = rng.integers(-(1 << 31), 1 << 31, n).astype(np.int32)
    small = rng.integers(-3, 4, n).astype(np.int32)
    f32 = rng.standard_normal(n).astype(np.float32)
    f32[:9] = specials32()
    f64 = rng.standard_normal(n)
    f64[:9] = specials64()
    words = ["", "a", "parquet", "page", "decode", "MI355X", "dictionary", "delta", "byte-stream-split", "été"]
    strs = [words[int(i)] + str(int(j)) for i, j in zip(rng.integers(0, len(words), n), rng.integers(0, 40, n))]
    sorted_strs = sorted("key/%05d/%s" % (int(i), words[int(i) % len(words)]) for i in rng.integers(0, 3000, n))
    yield "delta_v1", dict(data_page_version="1.0", compression="NONE", use_dictionary=False, data_page_size=8192,
                           column_encoding={"ts": "DELTA_BINARY_PACKED", "ts_jumps": "DELTA_BINARY_PACKED", "i32": "DELTA_BINARY_PACKED",
                                            "small": "DELTA_BINARY_PACKED", "ts_opt": "DELTA_BINARY_PACKED"}), \
        pa.table({"ts": ts, "ts_jumps": ts_jumps, "i32": i32, "small": small,
                  "ts_opt": with_nulls(ts, rng, 0.15, pa.int64())})
    yield "delta_wrap", dict(data_page_version="2.0", compression="NONE", use_dictionary=False,
                             column_encoding={"wrap": "DELTA_BINARY_PACKED", "one": "DELTA_BINARY_PACKED"}), \
        pa.table({"wrap": wrap, "one": np.full(300, 7, np.int64)})
    yield "bss", dict(data_page_version="1.0", compression="NONE", use_dictionary=False, data_page_size=8192,
                      column_encoding={"f32": "BYTE_STREAM_SPLIT", "f64": "BYTE_STREAM_SPLIT", "f64_opt": "BYTE_STREAM_SPLIT",
                                       "i32": "BYTE_STREAM_SPLIT", "i64": "BYTE_STREAM_SPLIT"}), \
        pa.table({"f32": f32, "f64": f64, "f64_opt": with_nulls(f64, rng, 0.3, pa.float64()), "i32": i32, "i64": ts})
    dict_i32 = rng.integers(0, 900, n).astype(np.int32) * 1001
    dict_i64 = ts[rng.integers(0, 300, n)]
    dict_f64 = f64[rng.integers(0, 50, n)]
    yield "dict_v2", dict(data_page_version="2.0", compression="NONE", use_dictionary=True, data_page_size=4096), \
        pa.table({"i32": dict_i32, "i64": dict_i64, "f64": dict_f64, "str": strs,
                  "i32_opt": with_nulls(dict_i32, rng, 0.2, pa.int32()),
                  "str_opt": with_nulls(strs, rng, 0.4, pa.string())})
    yield "dict_v1", dict(data_page_version="1.0", compression="NONE", use_dictionary=True, data_page_size=4096), \
        pa.table({"i64": dict_i64, "str_opt": with_nulls(strs, rng, 0.4, pa.string()),
                  "runs": np.repeat(rng.integers(0, 5, n // 50), 50).astype(np.int32)})
    bools = rng.random(n) < 0.3
    plain = {"i32": i32, "i64": ts, "f32": f32, "f64": f64, "bool": bools, "str": strs,
             "i32_opt": with_nulls(i32, rng, 0.25, pa.int32()), "f64_opt": with_nulls(f64, rng, 0.05, pa.float64()),
             "bool_opt": with_nulls(bools, rng, 0.5, pa.bool_()), "str_opt": with_nulls(strs, rng, 0.1, pa.string()),
             "all_null": pa.array([None] * n, type=pa.int32()),
             "flba": pa.array([bytes(rng.integers(0, 256, 5, dtype=np.uint8)) for _ in range(n)], type=pa.binary(5))}
    yield "plain_v1", dict(data_page_version="1.0", compression="NONE", use_dictionary=False, data_page_size=8192), pa.table(plain)
    yield "plain_v2", dict(data_page_version="2.0", compression="NONE", use_dictionary=False, data_page_size=8192), pa.table(plain)
    lists = pa.array([None if r < 0.1 else [int(x) for x in rng.integers(0, 100, int(rng.geometric(0.33)) - 1)]
                      for r in rng.random(1200)], type=pa.list_(pa.int32()))
    yield "nested_v1", dict(data_page_version="1.0", compression="NONE", use_dictionary=False, data_page_size=4096), \
        pa.table({"xs": lists})
    yield "nested_v2", dict(data_page_version="2.0", compression="NONE", use_dictionary=True, data_page_size=4096), \
        pa.table({"xs": lists})
    yield "delta_strings", dict(data_page_version="2.0", compression="NONE", use_dictionary=False, data_page_size=8192,
                                column_encoding={"dlba": "DELTA_LENGTH_BYTE_ARRAY", "dba": "DELTA_BYTE_ARRAY",
                                                 "dba_sorted": "DELTA_BYTE_ARRAY", "dlba_opt": "DELTA_LENGTH_BYTE_ARRAY"}), \
        pa.table({"dlba": strs, "dba": strs, "dba_sorted": sorted_strs, "dlba_opt": with_nulls(strs, rng, 0.2, pa.string())})
    comp = {"i64": ts, "str": strs, "f64_opt": with_nulls(f64, rng, 0.2, pa.float64()), "runs": np.repeat(rng.integers(0, 5, n // 50), 50).astype(np.int32)}
    for ver in ("1.0", "2.0"):  # page checksums (PageHeader.crc), plain and compressed
        yield "crc_v%s" % ver[0], dict(data_page_version=ver, compression={"i64": "NONE", "str": "SNAPPY", "f64_opt": "NONE", "runs": "ZSTD"},
                                      use_dictionary=["str", "runs"], data_page_size=4096, write_page_checksum=True), pa.table(comp)
    for codec in ("SNAPPY", "ZSTD", "LZ4_RAW"):
        for ver in ("1.0", "2.0"):
            yield "comp_%s_v%s" % (codec.lower(), ver[0]), dict(data_page_version=ver, compression=codec, use_dictionary=["str", "runs"],
                                                               data_page_size=8192), pa.table(comp)


def flatten(col):
    """Non-null leaf values in order and, per top-level row, the nesting pyarrow read."""
    arr = col.combine_chunks()
    if pa.types.is_list(arr.type):
        rows = arr.to_pylist()
        leaf = [x for r in rows if r is not None for x in r]
        return leaf, rows
    present = arr.is_valid().to_numpy(zero_copy_only=False).tolist()
    if pa.types.is_floating(arr.type) or pa.types.is_integer(arr.type):
        return arr.drop_null().to_numpy(zero_copy_only=False), present  # bit patterns survive (a Python float would quiet a signalling NaN)
    vals = arr.to_pylist()
    return [v for v in vals if v is not None], present


def b64(raw):
    return base64.b64encode(bytes(raw)).decode()


def encode_values(values, ptype):
    """Exact and compact: base64 of the little-endian array (bit patterns for floats); for byte arrays
    the lengths (int32) and the concatenated bytes."""
    if ptype == "FLOAT":
        return dict(dtype="<u4", data=b64(np.array(values, np.float32).view(np.uint32).tobytes()))
    if ptype == "DOUBLE":
        return dict(dtype="<u8", data=b64(np.array(values, np.float64).view(np.uint64).tobytes()))
    if ptype in ("BYTE_ARRAY", "FIXED_LEN_BYTE_ARRAY"):
        raw = [(v.encode("utf-8") if isinstance(v, str) else bytes(v)) for v in values]
        return dict(dtype="bytes", lengths=b64(np.array([len(r) for r in raw], np.int32).tobytes()), data=b64(b"".join(raw)))
    if ptype == "BOOLEAN":
        return dict(dtype="u1", data=b64(np.array(values, np.uint8).tobytes()))
    if ptype == "INT32":
        return dict(dtype="<i4", data=b64(np.array(values, np.int32).tobytes()))
    return dict(dtype="<i8", data=b64(np.array(values, np.int64).tobytes()))


def main():
    os.makedirs(TMP, exist_ok=True)
    fixtures = []
    for name, opts, table in tables():
        path = os.path.join(TMP, name + ".parquet")
        pq.write_table(table, path, write_statistics=False, **opts)
        buf = open(path, "rb").read()
        pf = pq.ParquetFile(path)
        md = pf.metadata
        assert md.num_row_groups == 1
        back = pf.read()
        g = md.row_group(0)
        for ci in range(md.num_columns):
            cc = g.column(ci)
            col = pf.schema.column(ci)
            start = cc.data_page_offset if cc.dictionary_page_offset is None else min(cc.dictionary_page_offset, cc.data_page_offset)
            chunk = buf[start:start + cc.total_compressed_size]
            top = cc.path_in_schema.split(".")[0]
            leaf, rows = flatten(back.column(top))
            fx = dict(file=name, column=cc.path_in_schema, physical_type=cc.physical_type, type_length=col.length if cc.physical_type == "FIXED_LEN_BYTE_ARRAY" else 0,
                      max_def=col.max_definition_level, max_rep=col.max_repetition_level, codec=cc.compression,
                      num_values=cc.num_values, encodings=list(cc.encodings), data_page_version=opts["data_page_version"],
                      chunk=base64.b64encode(chunk).decode(), values=encode_values(leaf, cc.physical_type))
            if col.max_repetition_level:
                # per row: -1 = null list, else its length (the leaf values are all non-null here)
                fx["row_lengths"] = b64(np.array([-1 if r is None else len(r) for r in rows], np.int32).tobytes())
            elif col.max_definition_level:
                fx["present"] = b64(np.packbits(np.array(rows, np.uint8), bitorder="little").tobytes())
            fixtures.append(fx)
    out = os.path.join(HERE, "pyarrow_chunks.json")
    with open(out, "w") as f:
        json.dump(dict(writer="pyarrow " + pa.__version__, fixtures=fixtures), f, separators=(",", ":"))
    print("wrote", out, os.path.getsize(out), "bytes,", len(fixtures), "column chunks")
    for fx in fixtures:
        print(" ", fx["file"], fx["column"], fx["physical_type"], fx["codec"], fx["encodings"], len(base64.b64decode(fx["chunk"])))


if __name__ == "__main__":
    sys.exit(main())
