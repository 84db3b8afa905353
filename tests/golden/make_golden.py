#!/usr/bin/env python3
"""Cuts level streams, dictionary-id pages and PLAIN value pages out of the .parquet files that parquet-java itself wrote
(the reference's own test resources) and stores them, with their expected decoded values, in
``java_pages.json``.  Run in the build container, where /root/reference exists:

    python tests/golden/make_golden.py

The fixtures are DATA (page bytes written by parquet-mr 1.9.0 / 1.13.0-SNAPSHOT and the values they
hold); nothing of the reference's source is copied.  Expected values come from a small pure-Python
reading of the format done here (independent of the C oracle and of the HIP kernels) and are
cross-checked against pyarrow's reading of the same files: dictionary-decoded strings must equal
pyarrow's column, definition levels must agree with pyarrow's null structure.
"""
import json
import os
import struct
import sys

import pyarrow.parquet as pq

REF = "/root/reference"
FILES = [
    "parquet-hadoop/src/test/resources/test-append_1.parquet",
    "parquet-hadoop/src/test/resources/test-append_2.parquet",
    "parquet-hadoop/src/test/resources/test-file-with-no-column-indexes-1.parquet",
    "parquet-avro/src/test/resources/strings-2.parquet",
]
PLAIN_FMT = {"INT32": ("i", 4), "INT64": ("q", 8), "DOUBLE": ("d", 8), "FLOAT": ("f", 4)}
ENCODINGS = {0: "PLAIN", 2: "PLAIN_DICTIONARY", 3: "RLE", 4: "BIT_PACKED", 5: "DELTA_BINARY_PACKED",
             8: "RLE_DICTIONARY", 9: "BYTE_STREAM_SPLIT"}


# ---- thrift compact protocol, just enough for parquet's PageHeader -------------------------------
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


def read_page_header(buf, pos):
    c = Compact(buf, pos)
    h = c.struct()
    return h, c.p


# ---- independent pure-Python readers (format spec level) -----------------------------------------
def py_rle_hybrid(data, w, n):
    out, p = [], 0
    while len(out) < n:
        h = s = 0
        while True:
            x = data[p]
            p += 1
            h |= (x & 0x7F) << s
            s += 7
            if not x & 0x80:
                break
        if h & 1:
            groups = h >> 1
            chunk = data[p:p + groups * w]
            p += len(chunk)
            big = int.from_bytes(bytes(chunk) + b"\0" * (groups * w - len(chunk)), "little")
            for i in range(groups * 8):
                out.append((big >> (i * w)) & ((1 << w) - 1))
        else:
            nb = (w + 7) // 8
            v = int.from_bytes(bytes(data[p:p + nb]), "little")
            p += nb
            out += [v] * (h >> 1)
    return out[:n]


def py_bitpacked_msb(data, w, n):
    big = int.from_bytes(bytes(data), "big") if len(data) else 0
    total = len(data) * 8
    return [(big >> (total - (i + 1) * w)) & ((1 << w) - 1) if w else 0 for i in range(n)]


def width_of(max_level):
    return max_level.bit_length()


def main():
    fixtures = []
    for rel in FILES:
        path = os.path.join(REF, rel)
        buf = open(path, "rb").read()
        pf = pq.ParquetFile(path)
        table = pf.read()
        md = pf.metadata
        row0 = 0
        for rg in range(md.num_row_groups):
            g = md.row_group(rg)
            for ci in range(md.num_columns):
                cc = g.column(ci)
                col = pf.schema.column(ci)
                max_def, max_rep = col.max_definition_level, col.max_repetition_level
                start = cc.data_page_offset if cc.dictionary_page_offset is None else min(cc.dictionary_page_offset, cc.data_page_offset)
                pos, end = start, start + cc.total_compressed_size
                dictionary = None
                page_no = 0
                decoded_strings = []
                def_levels_all = []
                plain_values = []
                while pos < end:
                    h, body = read_page_header(buf, pos)
                    size = h[3]
                    payload = buf[body:body + size]
                    pos = body + size
                    if h[1] == 2:  # DICTIONARY_PAGE: PLAIN BYTE_ARRAY entries [len LE32][bytes]
                        n = h[7][1]
                        dictionary_page = bytes(payload)
                        dictionary, q = [], 0
                        for _ in range(n):
                            ln = struct.unpack_from("<i", payload, q)[0]
                            dictionary.append(bytes(payload[q + 4:q + 4 + ln]))
                            q += 4 + ln
                        continue
                    assert h[1] == 0, "only DataPage V1 in these files"
                    dh = h[5]
                    n, enc, def_enc, rep_enc = dh[1], ENCODINGS[dh[2]], ENCODINGS[dh[3]], ENCODINGS[dh[4]]
                    q = 0
                    base = dict(file=os.path.basename(rel), row_group=rg, column=cc.path_in_schema, page=page_no,
                                created_by=md.created_by, num_values=n)
                    for kind, level_enc, max_level in (("rep", rep_enc, max_rep), ("def", def_enc, max_def)):
                        w = width_of(max_level)
                        if max_level == 0:
                            continue  # no bytes at all (ColumnReaderBase.readPageV1 / Encoding.getValuesReader)
                        if level_enc == "RLE":
                            ln = struct.unpack_from("<i", payload, q)[0]
                            sect = payload[q:q + 4 + ln]
                            levels = py_rle_hybrid(sect[4:], w, n)
                            fixtures.append(dict(base, kind=kind + "_levels", encoding="RLE", length_prefix=True,
                                                 bit_width=w, data=bytes(sect).hex(), expected=levels,
                                                 consumed=4 + ln))
                            q += 4 + ln
                        else:
                            ln = (n * w + 7) // 8
                            sect = payload[q:q + ln]
                            levels = py_bitpacked_msb(sect, w, n)
                            fixtures.append(dict(base, kind=kind + "_levels", encoding="BIT_PACKED", bit_width=w,
                                                 data=bytes(sect).hex(), expected=levels, consumed=ln))
                            q += ln
                        if kind == "def":
                            def_levels_all += levels
                    n_present = n if max_def == 0 else sum(1 for x in fixtures[-1]["expected"] if x == max_def)
                    if enc in ("PLAIN_DICTIONARY", "RLE_DICTIONARY"):
                        sect = payload[q:]
                        w = sect[0]
                        ids = py_rle_hybrid(sect[1:], w, n_present)
                        assert max(ids) < len(dictionary)
                        fixtures.append(dict(base, kind="dictionary_ids", encoding=enc, bit_width=w,
                                             data=bytes(sect).hex(), expected=ids, num_values=n_present,
                                             dictionary=[d.decode("latin-1") for d in dictionary],
                                             dictionary_page=dictionary_page.hex()))
                        decoded_strings += [dictionary[i] for i in ids]
                    elif enc == "PLAIN" and cc.physical_type in PLAIN_FMT:
                        # PLAIN fixed-width values as parquet-java wrote them: n_present little-endian values
                        fmt, size = PLAIN_FMT[cc.physical_type]
                        sect = payload[q:]
                        vals = list(struct.unpack_from("<%d%s" % (n_present, fmt), sect, 0))
                        fixtures.append(dict(base, kind="plain_values", encoding="PLAIN", physical_type=cc.physical_type,
                                             elem_size=size, data=bytes(sect).hex(), num_values=n_present,
                                             expected_hex=bytes(sect[:n_present * size]).hex()))
                        plain_values += vals
                    elif enc == "PLAIN" and cc.physical_type == "BYTE_ARRAY":
                        # PLAIN BYTE_ARRAY values as parquet-java wrote them: [length: 4 bytes LE][bytes] ...
                        sect = payload[q:]
                        vals, at = [], 0
                        for _ in range(n_present):
                            ln = struct.unpack_from("<i", sect, at)[0]
                            vals.append(bytes(sect[at + 4:at + 4 + ln]))
                            at += 4 + ln
                        assert at == len(sect)
                        fixtures.append(dict(base, kind="plain_byte_array", encoding="PLAIN", physical_type="BYTE_ARRAY",
                                             data=bytes(sect).hex(), num_values=n_present,
                                             expected=[v.decode("latin-1") for v in vals]))
                        decoded_strings += vals  # compared with pyarrow's reading of the column below
                    page_no += 1
                # ---- cross-checks against pyarrow's reading of the same chunk
                name = cc.path_in_schema
                if decoded_strings and max_rep == 0:
                    arr = table.column(name.split(".")[0]).slice(row0, g.num_rows).to_pylist()
                    got = [s.encode("latin-1") if isinstance(s, str) else s for s in arr if s is not None]
                    assert got == decoded_strings, name
                if def_levels_all and max_rep == 0 and "." not in name:
                    arr = table.column(name).slice(row0, g.num_rows).to_pylist()
                    assert [1 if v is not None else 0 for v in arr] == def_levels_all, name
                if def_levels_all and name.startswith("location."):
                    arr = table.column("location").slice(row0, g.num_rows).to_pylist()
                    f = name.split(".")[1]
                    exp = [0 if v is None else (1 if v[f] is None else 2) for v in arr]
                    assert exp == def_levels_all, name
                if plain_values and max_rep == 0:
                    if "." not in name:
                        arr = table.column(name).slice(row0, g.num_rows).to_pylist()
                    else:
                        top, f = name.split(".")
                        arr = [None if v is None else v[f] for v in table.column(top).slice(row0, g.num_rows).to_pylist()]
                    assert [v for v in arr if v is not None] == plain_values, name
                if decoded_strings and name == "phoneNumbers.phone.kind":
                    arr = table.column("phoneNumbers").slice(row0, g.num_rows).to_pylist()
                    flat = [p["kind"].encode() for v in arr if v is not None and v["phone"] is not None
                            for p in v["phone"] if p["kind"] is not None]
                    assert flat == decoded_strings, name
            row0 += g.num_rows
    # whole column chunks of the same files (page headers included), for the page-header walk
    import base64
    chunks = []
    for rel in FILES:
        path = os.path.join(REF, rel)
        buf = open(path, "rb").read()
        pf = pq.ParquetFile(path)
        md = pf.metadata
        for rg in range(md.num_row_groups):
            g = md.row_group(rg)
            for ci in range(md.num_columns):
                cc = g.column(ci)
                col = pf.schema.column(ci)
                start = cc.data_page_offset if cc.dictionary_page_offset is None else min(cc.dictionary_page_offset, cc.data_page_offset)
                chunks.append(dict(file=os.path.basename(rel), row_group=rg, column=cc.path_in_schema, created_by=md.created_by,
                                   physical_type=cc.physical_type, max_def=col.max_definition_level, max_rep=col.max_repetition_level,
                                   codec=cc.compression, num_values=cc.num_values, has_dictionary_page=cc.has_dictionary_page,
                                   encodings=list(cc.encodings), chunk=base64.b64encode(buf[start:start + cc.total_compressed_size]).decode()))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "java_chunks.json")
    with open(out, "w") as f:
        json.dump(dict(source="column chunks of parquet-java test resources", fixtures=chunks), f, separators=(",", ":"))
    print("wrote", out, os.path.getsize(out), "bytes;", len(chunks), "column chunks")
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "java_pages.json")
    with open(out, "w") as f:
        json.dump(dict(source="parquet-java test resources: " + ", ".join(FILES), fixtures=fixtures), f, separators=(",", ":"))
    kinds = {}
    for fx in fixtures:
        kinds[(fx["kind"], fx["encoding"])] = kinds.get((fx["kind"], fx["encoding"]), 0) + 1
    print("wrote", out, os.path.getsize(out), "bytes;", kinds)


if __name__ == "__main__":
    sys.exit(main())
