"""The page-header walk of the library (pqd_chunk_walk: thrift compact PageHeader parse, CRC-32 check;
host code, no GPU needed) against an independent pure-Python walk (tests/pagefile.py) on every column
chunk parquet-java wrote into the reference's test resources and every chunk of the pyarrow fixtures;
then whole chunks through the device: sections -> levels -> validity -> values, against the values
pyarrow reads."""
import base64
import ctypes as C
import json
import os
import zlib

import numpy as np
import pytest

import pagefile as pf

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "java_chunks.json")) as _f:
    JAVA = json.load(_f)["fixtures"]
ARROW = pf.load_fixtures() + pf.load_fixtures("pyarrow_gzip_chunks.json")
ENC_NUM = {v: k for k, v in pf.ENCODINGS.items()}


def walk(pq, chunk, value_count, verify_crc=1, cap=4096):
    buf = np.frombuffer(chunk, np.uint8)
    pages = (pq.ChunkPage * cap)()
    n = C.c_uint32(0)
    st = pq.lib.pqd_chunk_walk(buf.ctypes.data_as(C.c_void_p) if buf.size else None, buf.size, value_count, verify_crc, pages, cap, C.byref(n))
    return st, [pages[i] for i in range(min(n.value, cap))], n.value


@pytest.mark.parametrize("fx", JAVA + ARROW, ids=lambda fx: fx["file"] + "." + fx["column"] + str(fx.get("row_group", "")))
def test_walk_matches_the_independent_reader(fx):
    import parquet_java_amd as pq
    chunk = base64.b64decode(fx["chunk"])
    want = pf.walk_chunk(chunk)
    st, got, n = walk(pq, chunk, fx["num_values"])
    assert st == 0 and n == len(want)
    for g, w in zip(got, want):
        assert (g.header_offset, g.body_offset, g.compressed_size, g.uncompressed_size) == (w.header_at, w.body_at, w.compressed_size, w.uncompressed_size)
        assert g.type == w.type and g.num_values == w.num_values and g.encoding == ENC_NUM[w.encoding]
        if w.type == pf.DATA_PAGE:
            assert g.def_encoding == ENC_NUM[w.def_encoding] and g.rep_encoding == ENC_NUM[w.rep_encoding]
        if w.type == pf.DATA_PAGE_V2:
            assert (g.num_nulls, g.num_rows, g.def_levels_len, g.rep_levels_len, bool(g.is_compressed)) == \
                   (w.num_nulls, w.num_rows, w.def_len, w.rep_len, bool(w.is_compressed))
        assert bool(g.has_crc) == (w.crc is not None)
        if w.crc is not None:
            assert g.crc == w.crc & 0xFFFFFFFF == zlib.crc32(w.body)
    # the same walk without a value count stops at the end of the bytes
    st2, got2, n2 = walk(pq, chunk, -1)
    assert st2 == 0 and n2 == n
    assert sum(g.num_values for g in got if g.type != pf.DICTIONARY_PAGE) == fx["num_values"]


def test_walk_failures():
    import parquet_java_amd as pq
    fx = next(f for f in ARROW if f["file"] == "crc_v1" and f["column"] == "i64")
    chunk = bytearray(base64.b64decode(fx["chunk"]))
    pages = pf.walk_chunk(bytes(chunk))
    assert len(pages) >= 2 and pages[0].crc is not None
    assert walk(pq, bytes(chunk), fx["num_values"])[0] == 0
    assert pq.lib.pqd_crc32(0, bytes(chunk), len(chunk)) == zlib.crc32(bytes(chunk))
    # one flipped bit in a page body: "could not verify page integrity, CRC checksum verification failed"
    bad = bytearray(chunk)
    bad[pages[1].body_at + 5] ^= 0x10
    assert walk(pq, bytes(bad), fx["num_values"])[0] == pq.DECODE
    assert walk(pq, bytes(bad), fx["num_values"], verify_crc=0)[0] == 0  # not checked unless asked for
    # a value count the pages do not add up to: "Expected N values in column chunk ... but got M"
    assert walk(pq, bytes(chunk), fx["num_values"] - 1)[0] == pq.DECODE
    # more values expected than the chunk holds: the next header is read past the end
    assert walk(pq, bytes(chunk), fx["num_values"] + 1)[0] == pq.EOF
    # the chunk cut inside a page, inside a header
    assert walk(pq, bytes(chunk[:pages[1].body_at + 10]), fx["num_values"])[0] == pq.EOF
    assert walk(pq, bytes(chunk[:pages[1].header_at + 3]), fx["num_values"])[0] == pq.EOF
    # room for fewer pages than there are: the count needed comes back
    st, got, n = walk(pq, bytes(chunk), fx["num_values"], cap=1)
    assert st == pq.BAD_ARG and n == len(pages) and len(got) == 1
    # two dictionary pages: "more than one dictionary page in column"
    dfx = next(f for f in ARROW if f["file"] == "dict_v1" and f["column"] == "i64")
    dchunk = base64.b64decode(dfx["chunk"])
    dpages = pf.walk_chunk(dchunk)
    assert dpages[0].type == pf.DICTIONARY_PAGE
    twice = dchunk[:dpages[1].header_at] + dchunk[:dpages[1].header_at] + dchunk[dpages[1].header_at:]
    assert walk(pq, twice, dfx["num_values"])[0] == pq.DECODE
    # garbage never walks off the buffer
    rng = np.random.default_rng(3)
    for _ in range(300):
        junk = rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8).tobytes()
        assert walk(pq, junk, int(rng.integers(-1, 50)))[0] in (0, pq.EOF, pq.DECODE)
    for _ in range(300):
        bad = bytearray(chunk[:400])
        for at in rng.integers(0, 40, 3):
            bad[at] = int(rng.integers(0, 256))
        assert walk(pq, bytes(bad), fx["num_values"])[0] in (0, pq.EOF, pq.DECODE)


# ------------------------------------------------------------------------------- whole chunks on the device
PHYS = {"BOOLEAN": 0, "INT32": 1, "INT64": 2, "INT96": 3, "FLOAT": 4, "DOUBLE": 5, "BYTE_ARRAY": 6, "FIXED_LEN_BYTE_ARRAY": 7}
CODECS = {"UNCOMPRESSED": 0, "SNAPPY": 1, "LZ4": 7, "ZSTD": 6, "GZIP": 2}  # (pyarrow's metadata calls LZ4_RAW "LZ4")
WHOLE = [fx for fx in ARROW if fx["codec"] in CODECS and fx["max_rep"] == 0 and
         fx["physical_type"] in ("INT32", "INT64", "FLOAT", "DOUBLE", "BOOLEAN", "FIXED_LEN_BYTE_ARRAY", "BYTE_ARRAY")]


def test_whole_chunk_cases_cover_the_byte_array_encodings():
    kinds = {tuple(sorted(set(fx["encodings"]) - {"RLE"})) for fx in WHOLE if fx["physical_type"] == "BYTE_ARRAY"}
    assert {("PLAIN",), ("PLAIN", "RLE_DICTIONARY"), ("DELTA_LENGTH_BYTE_ARRAY",), ("DELTA_BYTE_ARRAY",)} <= kinds


@pytest.mark.gpu
@pytest.mark.parametrize("fx", WHOLE, ids=lambda fx: fx["file"] + "." + fx["column"])
def test_gpu_whole_chunk_from_one_blob(orc, fx):
    """The chunk's bytes go to the device once; page table from pqd_chunk_walk; sections, level decode,
    validity, value counts and value decode are device launches over all pages of the chunk."""
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    chunk = base64.b64decode(fx["chunk"])
    st, pages, n = walk(pq, chunk, fx["num_values"])
    assert st == 0
    d_chunk = gh.to_dev(np.concatenate([np.frombuffer(chunk, np.uint8), np.zeros(64, np.uint8)]))
    binary = fx["physical_type"] == "BYTE_ARRAY"
    elem = {"INT32": 4, "INT64": 8, "FLOAT": 4, "DOUBLE": 8, "BOOLEAN": 1, "FIXED_LEN_BYTE_ARRAY": fx["type_length"], "BYTE_ARRAY": 8}[fx["physical_type"]]
    d_bodies = d_chunk
    if CODECS[fx["codec"]]:
        # compressed pages: every body is decompressed on the device into one buffer, which then is the chunk
        cap = sum((p.uncompressed_size + 15) & ~15 for p in pages) + 64
        d_bodies = torch.full((cap,), 0xA5, dtype=torch.uint8, device=gh.dev())
        table_in, table_out = (pq.ChunkPage * len(pages))(*pages), (pq.ChunkPage * len(pages))()
        d_cd = torch.zeros(2 * len(pages) * 32, dtype=torch.uint8, device=gh.dev())
        d_cr = torch.zeros(2 * len(pages) * 16, dtype=torch.uint8, device=gh.dev())
        pq.check(pq.lib.pqd_chunk_decompress_dev(d_chunk.data_ptr(), table_in, len(pages), CODECS[fx["codec"]], d_bodies.data_ptr(),
                                                 cap - 64, table_out, d_cd.data_ptr(), d_cr.data_ptr(), None))
        torch.cuda.synchronize()
        assert (pq.results_from_numpy(d_cr.cpu().numpy())[0] == 0).all()
        pages = list(table_out)
        assert all(p.is_compressed == 0 and p.compressed_size == p.uncompressed_size for p in pages)
    dict_page = [p for p in pages if p.type == pq.PAGE_DICTIONARY]
    data_pages = [p for p in pages if p.type != pq.PAGE_DICTIONARY]
    d_dict, dict_len, handle, info = None, 0, C.c_void_p(), pq.DictInfo()
    if dict_page:
        p = dict_page[0]
        body = d_bodies.cpu().numpy()[p.body_offset:p.body_offset + p.compressed_size]
        d_dict = gh.to_dev(np.concatenate([body, np.zeros(16, np.uint8)]))  # PLAIN fixed-width dictionary page = the entries
        dict_len = p.num_values
        if binary:  # entries = views into the dictionary page's bytes (pqd_dict_create walks the page)
            body = np.ascontiguousarray(body)
            pq.check(pq.lib.pqd_dict_create(PHYS["BYTE_ARRAY"], 0, body.ctypes.data, body.size, p.num_values, C.byref(handle)))
            pq.check(pq.lib.pqd_dict_get(handle, C.byref(info)))
    table = (pq.ChunkPage * len(data_pages))(*data_pages)
    d_table = gh.to_dev(np.frombuffer(bytes(table), np.uint8).copy())
    n_pages = len(data_pages)
    level_base = np.cumsum([0] + [p.num_values for p in data_pages]).astype(np.uint32)
    total = int(level_base[-1])
    d_base = gh.to_dev(level_base.view(np.uint8))
    d_def = torch.zeros(max(total, 1), dtype=torch.int32, device=gh.dev())
    out_elem = 4 if (fx["physical_type"] == "BOOLEAN" and "RLE" in fx["encodings"] and fx["data_page_version"] == "2.0") else elem
    d_val = torch.zeros(max(total, 1) * max(out_elem, elem) + 64, dtype=torch.uint8, device=gh.dev())
    meta = pq.ColumnMeta(PHYS[fx["physical_type"]], fx["type_length"], fx["max_def"], 0)
    outs = pq.ChunkOutputs(d_bodies.data_ptr(), 0, d_def.data_ptr(), d_val.data_ptr(), d_base.data_ptr(), d_base.data_ptr(),
                           0 if d_dict is None else (info.entries if binary else d_dict.data_ptr()), dict_len, 0)
    descs = [torch.zeros(n_pages * 40, dtype=torch.uint8, device=gh.dev()) for _ in range(3)]
    res = [torch.zeros(n_pages * 16, dtype=torch.uint8, device=gh.dev()) for _ in range(4)]
    pq.check(pq.lib.pqd_chunk_sections_dev(d_table.data_ptr(), n_pages, C.byref(meta), C.byref(outs), descs[0].data_ptr(),
                                           descs[1].data_ptr(), descs[2].data_ptr(), res[0].data_ptr(), None))
    torch.cuda.synchronize()
    assert (pq.results_from_numpy(res[0].cpu().numpy())[0] == 0).all()
    host_descs = [d.cpu().numpy().view(np.uint8).reshape(n_pages, 40) for d in descs]
    # definition levels: one launch per level encoding (RLE here)
    if fx["max_def"]:
        pq.decode_pages_dev(descs[1].data_ptr(), res[1].data_ptr(), n_pages, pq.ENC_RLE, 4, None)
    # validity + non-null counts per page
    vdesc = (pq.ValidityDesc * n_pages)()
    d_validity = torch.zeros((total + 7) // 8 + n_pages + 8, dtype=torch.uint8, device=gh.dev())
    vpos = 0
    vstarts = []
    for i, p in enumerate(data_pages):
        vdesc[i].def_levels = d_def.data_ptr() + 4 * int(level_base[i]) if fx["max_def"] else 0
        vdesc[i].validity = d_validity.data_ptr() + vpos
        vdesc[i].n_levels = p.num_values
        vdesc[i].max_def = fx["max_def"]
        vstarts.append(vpos)
        vpos += (p.num_values + 7) // 8
    d_vdesc = gh.to_dev(np.frombuffer(bytes(vdesc), np.uint8).copy())
    pq.check(pq.lib.pqd_def_levels_to_validity_dev(d_vdesc.data_ptr(), res[2].data_ptr(), n_pages, None))
    pq.check(pq.lib.pqd_chunk_set_value_counts_dev(descs[2].data_ptr(), res[2].data_ptr(), n_pages, None))
    torch.cuda.synchronize()
    # values: one launch per encoding class among the pages (a dictionary chunk may end with PLAIN pages)
    val_descs = descs[2].cpu().numpy().reshape(n_pages, 40)
    encs = val_descs[:, 36]
    assert (encs != 0).all()
    order = np.argsort(encs, kind="stable")
    counts = res[2].cpu().numpy().view(np.uint32).reshape(-1, 4)[:, 3]
    if binary:
        # BYTE_ARRAY values -> offsets + bytes per page: dictionary pages in one launch (ids, dictionary entries and
        # bytes in one pass), the PLAIN / DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY pages in another
        want = pf.fixture_values(fx)
        room = sum(len(v) for v in want) + 64
        d_offs = torch.full((total + n_pages + 8,), -7, dtype=torch.int32, device=gh.dev())
        d_bytes = torch.zeros(n_pages * room, dtype=torch.uint8, device=gh.dev())
        fields = val_descs.view(np.uint64).reshape(n_pages, 5)[:, 0], val_descs.view(np.uint32).reshape(n_pages, 10)
        got_status = np.zeros(n_pages, np.int64)
        got_written = np.zeros(n_pages, np.int64)
        dict_pages = [i for i in range(n_pages) if encs[i] == pq.ENC_RLE_DICTIONARY]
        other_pages = [i for i in range(n_pages) if encs[i] != pq.ENC_RLE_DICTIONARY]
        assert all(encs[i] in (8, 9, 10) and val_descs[i, 37] == 0 for i in other_pages)
        if dict_pages:
            bins = (pq.BinaryPageDesc * len(dict_pages))()
            for k, i in enumerate(dict_pages):
                assert val_descs[i, 37] == 8 and fields[1][i, 8] == dict_len
                bins[k].dict_bytes, bins[k].dict_bytes_len = info.bytes, info.bytes_len
                bins[k].offsets = d_offs.data_ptr() + 4 * (int(level_base[i]) + i)
                bins[k].bytes, bins[k].bytes_capacity = d_bytes.data_ptr() + i * room, room
            d_bins = gh.to_dev(np.frombuffer(bytes(bins), np.uint8).copy())
            d_pages = gh.to_dev(val_descs[dict_pages].reshape(-1))
            r2 = [torch.zeros(len(dict_pages) * 16, dtype=torch.uint8, device=gh.dev()) for _ in range(2)]
            pq.check(pq.lib.pqd_decode_binary_pages_dev(handle, d_pages.data_ptr(), d_bins.data_ptr(), r2[0].data_ptr(), r2[1].data_ptr(),
                                                        len(dict_pages), None))
            torch.cuda.synchronize()
            for r in r2:
                st, wr, _ = pq.results_from_numpy(r.cpu().numpy())
                got_status[dict_pages] |= st
                got_written[dict_pages] = wr
        if other_pages:
            dd = (pq.DeltaBytesDesc * len(other_pages))()
            works = []
            for k, i in enumerate(other_pages):
                nv = int(fields[1][i, 7])
                works.append(torch.zeros(pq._native.delta_bytes_work(nv) + 16, dtype=torch.uint8, device=gh.dev()))
                dd[k].data, dd[k].data_len, dd[k].num_values, dd[k].encoding = int(fields[0][i]), int(fields[1][i, 6]), nv, int(encs[i])
                dd[k].offsets = d_offs.data_ptr() + 4 * (int(level_base[i]) + i)
                dd[k].bytes, dd[k].bytes_capacity = d_bytes.data_ptr() + i * room, room
                dd[k].work = works[-1].data_ptr()
            d_dd = gh.to_dev(np.frombuffer(bytes(dd), np.uint8).copy())
            r3 = torch.zeros(len(other_pages) * 16, dtype=torch.uint8, device=gh.dev())
            pq.check(pq.lib.pqd_delta_byte_array_dev(d_dd.data_ptr(), r3.data_ptr(), len(other_pages), None))
            torch.cuda.synchronize()
            st, wr, _ = pq.results_from_numpy(r3.cpu().numpy())
            got_status[other_pages], got_written[other_pages] = st, wr
        assert (got_status == 0).all() and (got_written == counts).all()
        offs, data = d_offs.cpu().numpy(), d_bytes.cpu().numpy()
        got = []
        for i in range(n_pages):
            o = offs[int(level_base[i]) + i:int(level_base[i]) + i + int(counts[i]) + 1]
            page_bytes = data[i * room:(i + 1) * room]
            got += [page_bytes[a:b].tobytes() for a, b in zip(o[:-1], o[1:])]
        assert got == want
        if handle:
            pq.check(pq.lib.pqd_dict_destroy(handle))
        want_present = pf.fixture_present(fx)
        if want_present is not None:
            validity = d_validity.cpu().numpy()
            bits = np.concatenate([np.unpackbits(validity[s:s + (p.num_values + 7) // 8], bitorder="little")[:p.num_values]
                                   for s, p in zip(vstarts, data_pages)]).astype(bool)
            assert (bits == want_present).all()
        return
    sorted_descs = gh.to_dev(val_descs[order].reshape(-1))
    sorted_res = torch.zeros(n_pages * 16, dtype=torch.uint8, device=gh.dev())
    a = 0
    while a < n_pages:
        b = a
        while b < n_pages and encs[order[b]] == encs[order[a]]:
            b += 1
        pq.decode_pages_dev(sorted_descs.data_ptr() + 40 * a, sorted_res.data_ptr() + 16 * a, b - a, int(encs[order[a]]),
                            int(val_descs[order[a], 37]), None)
        a = b
    torch.cuda.synchronize()
    status, written, _ = pq.results_from_numpy(sorted_res.cpu().numpy())
    assert (status == 0).all()
    assert (written == counts[order]).all()
    # gather the dense values page by page and compare with pyarrow's
    val = d_val.cpu().numpy()
    got = []
    for i, p in enumerate(data_pages):
        e = int(val_descs[i, 37])
        piece = val[int(level_base[i]) * e:(int(level_base[i]) + int(counts[i])) * e]
        if fx["physical_type"] == "BOOLEAN" and e == 4:
            piece = piece.view(np.int32).astype(np.uint8)
        got.append(piece)
    got = np.concatenate(got) if got else np.zeros(0, np.uint8)
    v = pf.fixture_values(fx)
    want = np.frombuffer(b"".join(v), np.uint8) if isinstance(v, list) else v.view(np.uint8)
    assert got.size == want.size and (got == want).all()
    want_present = pf.fixture_present(fx)
    if want_present is not None:
        validity = d_validity.cpu().numpy()
        bits = np.concatenate([np.unpackbits(validity[s:s + (p.num_values + 7) // 8], bitorder="little")[:p.num_values]
                               for s, p in zip(vstarts, data_pages)]).astype(bool)
        assert (bits == want_present).all()
