"""Writes the GZIP fixtures: gzip_vectors.json (byte strings compressed by the zlib library: through pyarrow.Codec
at three levels, and through Python's zlib module for the forms a page writer's default settings do not produce:
fixed Huffman codes, stored blocks, several members, header fields) and pyarrow_gzip_chunks.json (column chunks of
GZIP-compressed parquet files written by pyarrow, V1 and V2 pages, in the format of pyarrow_chunks.json).  The
originals of the vectors are regenerated from seeds by tests/codec_inputs.py.  Run by hand:
    python tests/golden/make_gzip_golden.py"""
import base64
import gzip
import hashlib
import io
import json
import os
import sys
import zlib

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from codec_inputs import inputs  # noqa: E402
import make_pyarrow_golden as mpg  # noqa: E402


def gz(raw, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=31):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
    return c.compress(raw) + c.flush()


def with_header_fields(raw):
    """A member with FEXTRA, FNAME, FCOMMENT and FHCRC in its header (RFC 1952 section 2.3.1)."""
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = c.compress(raw) + c.flush()
    head = bytes([0x1F, 0x8B, 8, 4 | 8 | 16 | 2, 1, 2, 3, 4, 0, 3]) + (5).to_bytes(2, "little") + b"extra" + b"name.bin\0" + b"a comment\0"
    head += (zlib.crc32(head) & 0xFFFF).to_bytes(2, "little")
    return head + body + zlib.crc32(raw).to_bytes(4, "little") + (len(raw) & 0xFFFFFFFF).to_bytes(4, "little")


def main():
    vectors = []

    def add(name, raw, comp):
        assert gzip.GzipFile(fileobj=io.BytesIO(comp)).read() == raw
        vectors.append(dict(name=name, codec="GZIP", size=len(raw), sha256=hashlib.sha256(raw).hexdigest(),
                            compressed=base64.b64encode(comp).decode()))
        print(name, len(raw), "->", len(comp))

    for name, raw in inputs():
        small = len(raw) <= 20000  # (the large inputs in fewer forms: the file stays small)
        for level in (1, 6, 9) if small else (6,):
            comp = pa.Codec("gzip", compression_level=level).compress(raw, asbytes=True)
            add("%s_l%d" % (name, level), raw, comp)
        add(name + "_fixed", raw, gz(raw, 6, zlib.Z_FIXED))
        if small:
            add(name + "_stored", raw, gz(raw, 0))
            add(name + "_huffman_only", raw, gz(raw, 6, zlib.Z_HUFFMAN_ONLY))
        if len(raw) > 10:
            a, b, c = raw[:len(raw) // 3], raw[len(raw) // 3:len(raw) // 2], raw[len(raw) // 2:]
            add(name + "_three_members", raw, gz(a, 9) + gz(b, 0) + gz(c, 1, zlib.Z_FIXED))
        add(name + "_header_fields", raw, with_header_fields(raw))
    with open(os.path.join(HERE, "gzip_vectors.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_gzip_golden.py", pyarrow=pa.__version__, zlib=zlib.ZLIB_RUNTIME_VERSION, vectors=vectors), f)

    # ---- GZIP-compressed column chunks
    os.makedirs(mpg.TMP, exist_ok=True)
    rng = np.random.default_rng(20261112)
    n = 4000
    words = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "eta", "theta"]
    strs = pa.array([" ".join(words[int(k)] for k in rng.integers(0, 8, int(rng.integers(1, 6)))) for _ in range(n)])
    table = pa.table({"i64": pa.array(np.cumsum(rng.integers(0, 1000, n)).astype(np.int64)), "str": strs,
                      "f64_opt": mpg.with_nulls(rng.standard_normal(n), rng, 0.2, pa.float64()),
                      "runs": pa.array(np.repeat(rng.integers(0, 5, n // 50), 50).astype(np.int32)),
                      "plain_str": strs})
    fixtures = []
    for ver in ("1.0", "2.0"):
        name = "comp_gzip_v%s" % ver[0]
        opts = dict(data_page_version=ver, compression="GZIP", use_dictionary=["str", "runs"], data_page_size=8192)
        path = os.path.join(mpg.TMP, name + ".parquet")
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
            leaf, rows = mpg.flatten(back.column(cc.path_in_schema.split(".")[0]))
            fx = dict(file=name, column=cc.path_in_schema, physical_type=cc.physical_type, type_length=0,
                      max_def=col.max_definition_level, max_rep=col.max_repetition_level, codec=cc.compression,
                      num_values=cc.num_values, encodings=list(cc.encodings), data_page_version=ver,
                      chunk=base64.b64encode(chunk).decode(), values=mpg.encode_values(leaf, cc.physical_type))
            if col.max_definition_level:
                fx["present"] = mpg.b64(np.packbits(np.array(rows, np.uint8), bitorder="little").tobytes())
            fixtures.append(fx)
            print(" ", name, fx["column"], fx["physical_type"], fx["codec"], fx["encodings"], len(chunk))
    out = os.path.join(HERE, "pyarrow_gzip_chunks.json")
    with open(out, "w") as f:
        json.dump(dict(writer="pyarrow " + pa.__version__, fixtures=fixtures), f, separators=(",", ":"))
    print("wrote", out, os.path.getsize(out), "bytes,", len(fixtures), "column chunks")


if __name__ == "__main__":
    main()
