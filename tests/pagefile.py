"""Test-side reading of a column chunk: thrift compact ``PageHeader`` walk and the split of a data page
into repetition levels, definition levels and values (the layout ParquetFileReader.Chunk.readAllPages
and ColumnReaderBase.readPageV1 / readPageV2 consume: parquet-hadoop/.../ParquetFileReader.java:1973-2117,
parquet-column/.../impl/ColumnReaderBase.java:729-780).  Pure Python, independent of the C oracle and
of the library's own page-header walk, which the tests compare with it."""
import base64
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ENCODINGS = {0: "PLAIN", 2: "PLAIN_DICTIONARY", 3: "RLE", 4: "BIT_PACKED", 5: "DELTA_BINARY_PACKED",
             6: "DELTA_LENGTH_BYTE_ARRAY", 7: "DELTA_BYTE_ARRAY", 8: "RLE_DICTIONARY", 9: "BYTE_STREAM_SPLIT"}
DATA_PAGE, INDEX_PAGE, DICTIONARY_PAGE, DATA_PAGE_V2 = 0, 1, 2, 3
ELEM = {"INT32": 4, "INT64": 8, "FLOAT": 4, "DOUBLE": 8, "INT96": 12}


class Compact:
    def __init__(self, buf, pos):
        self.b, self.p = buf, pos

    def byte(self):
        v = self.b[self.p]
        self.p += 1
        return v

    def varint(self):
        v = s = 0
        while True:
            x = self.byte()
            v |= (x & 0x7F) << s
            s += 7
            if not x & 0x80:
                return v

    def zigzag(self):
        v = self.varint()
        return (v >> 1) ^ -(v & 1)

    def skip(self, t):
        if t in (1, 2):
            return
        if t == 3:
            self.byte()
        elif t in (4, 5, 6):
            self.zigzag()
        elif t == 7:
            self.p += 8
        elif t == 8:
            n = self.varint()
            self.p += n
        elif t in (9, 10):
            h = self.byte()
            n = h >> 4
            if n == 15:
                n = self.varint()
            for _ in range(n):
                self.skip(h & 15)
        elif t == 11:
            n = self.varint()
            if n:
                kv = self.byte()
                for _ in range(n):
                    self.skip(kv >> 4)
                    self.skip(kv & 15)
        elif t == 12:
            self.struct()
        else:
            raise ValueError("thrift type %d" % t)

    def struct(self):
        out, last = {}, 0
        while True:
            h = self.byte()
            if h == 0:
                return out
            delta, t = h >> 4, h & 15
            fid = last + delta if delta else self.zigzag()
            last = fid
            if t in (1, 2):
                out[fid] = t == 1
            elif t in (4, 5, 6):
                out[fid] = self.zigzag()
            elif t == 12:
                out[fid] = self.struct()
            else:
                self.skip(t)


class PageInfo:
    """One page of a chunk: header fields and the (still compressed) body."""

    def __init__(self, h, body, header_at, body_at):
        self.type = h[1]
        self.uncompressed_size, self.compressed_size = h[2], h[3]
        self.crc = h.get(4)
        self.body = body
        self.header_at, self.body_at = header_at, body_at
        self.num_values = self.encoding = None
        self.def_encoding = self.rep_encoding = None
        self.def_len = self.rep_len = 0
        self.num_nulls = self.num_rows = None
        self.is_compressed = True
        if self.type == DATA_PAGE:
            d = h[5]
            self.num_values, self.encoding = d[1], ENCODINGS[d[2]]
            self.def_encoding, self.rep_encoding = ENCODINGS[d[3]], ENCODINGS[d[4]]
        elif self.type == DICTIONARY_PAGE:
            d = h[7]
            self.num_values, self.encoding = d[1], ENCODINGS[d[2]]
        elif self.type == DATA_PAGE_V2:
            d = h[8]
            self.num_values, self.num_nulls, self.num_rows, self.encoding = d[1], d[2], d[3], ENCODINGS[d[4]]
            self.def_len, self.rep_len = d[5], d[6]
            self.is_compressed = d.get(7, True)


def walk_chunk(chunk):
    chunk = bytes(chunk)
    pages, pos = [], 0
    while pos < len(chunk):
        c = Compact(chunk, pos)
        h = c.struct()
        body = chunk[c.p:c.p + h[3]]
        assert len(body) == h[3], "page runs past the chunk"
        pages.append(PageInfo(h, body, pos, c.p))
        pos = c.p + h[3]
    return pages


def level_width(max_level):
    return int(max_level).bit_length()


def split_v1(body, max_rep, max_def, rep_encoding="RLE", def_encoding="RLE", num_values=0):
    """(rep section incl. its 4-byte length, def section likewise, values) of an UNCOMPRESSED V1 page body."""
    q = 0
    out = []
    for max_level, enc in ((max_rep, rep_encoding), (max_def, def_encoding)):
        if max_level == 0:
            out.append(b"")
            continue
        if enc == "RLE":
            ln = struct.unpack_from("<i", body, q)[0]
            out.append(body[q:q + 4 + ln])
            q += 4 + ln
        else:  # BIT_PACKED
            ln = (num_values * level_width(max_level) + 7) // 8
            out.append(body[q:q + ln])
            q += ln
    return out[0], out[1], body[q:]


def split_v2(page):
    """(rep bytes, def bytes, values bytes possibly compressed) of a V2 page body."""
    b = page.body
    return b[:page.rep_len], b[page.rep_len:page.rep_len + page.def_len], b[page.rep_len + page.def_len:]


def load_fixtures(name="pyarrow_chunks.json"):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)["fixtures"]


def fixture_chunk(fx):
    return base64.b64decode(fx["chunk"])


def fixture_values(fx):
    """numpy array of the expected non-null values (bit patterns for floats), or a list of bytes."""
    v = fx["values"]
    raw = base64.b64decode(v["data"])
    if v["dtype"] == "bytes":
        lens = np.frombuffer(base64.b64decode(v["lengths"]), "<i4")
        ends = np.cumsum(lens)
        return [raw[e - l:e] for e, l in zip(ends, lens)]
    return np.frombuffer(raw, v["dtype"]).copy()


def fixture_present(fx):
    """Per-row presence of an optional flat column (None for required columns)."""
    if "present" not in fx:
        return None
    bits = np.unpackbits(np.frombuffer(base64.b64decode(fx["present"]), np.uint8), bitorder="little")
    return bits[:fx["num_values"]].astype(bool)
