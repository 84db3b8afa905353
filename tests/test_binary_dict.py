"""BINARY / FIXED_LEN_BYTE_ARRAY dictionaries: the dictionary-page walk (host side of the C ABI, no
GPU needed), the oracle's restatement of it, and -- on the GPU -- the gather of a page's values into
offsets + bytes, all against the reference's behaviour
(PlainValuesDictionary.PlainBinaryDictionary, PlainValuesDictionary.java:59-135;
DictionaryValuesReader.readBytes, DictionaryValuesReader.java:76-83) and the dictionary pages that
parquet-java itself wrote (tests/golden/java_pages.json)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
INVALID = 0xFFFFFFFFFFFFFFFF


def java_dictionaries():
    with open(os.path.join(HERE, "golden", "java_pages.json")) as f:
        return [fx for fx in json.load(f)["fixtures"] if fx["kind"] == "dictionary_ids"]


def lib_views(page, n, fixed=0):
    import parquet_java_amd as pq
    page = np.ascontiguousarray(page, np.uint8)
    views = np.zeros(max(n, 1), np.uint64)
    st = pq.lib.pqd_binary_dict_views(page.ctypes.data_as(C.c_void_p) if page.size else None, page.size, n, fixed,
                                      views.ctypes.data_as(C.c_void_p))
    return st, views[:n]


def random_dictionary(rng, n, max_len):
    return [rng.integers(0, 256, int(rng.integers(0, max_len + 1)), dtype=np.uint8).tobytes() for _ in range(n)]


# ------------------------------------------------------------------------------- host: dictionary walk
def test_views_of_java_written_dictionary_pages(orc):
    fx = java_dictionaries()
    assert len(fx) >= 10
    for f in fx:
        page = np.frombuffer(bytes.fromhex(f["dictionary_page"]), np.uint8)
        expect = [s.encode("latin-1") for s in f["dictionary"]]
        for st, views in (orc.binary_dict_views(page, len(expect)), lib_views(page, len(expect))):
            assert st == 0
            got = [page[int(v) & 0xFFFFFFFF:(int(v) & 0xFFFFFFFF) + (int(v) >> 32)].tobytes() for v in views]
            assert got == expect, f["column"]


def test_views_patterns_from_the_reference_tests(orc):
    # TestDictionary.testBinaryDictionary (TestDictionary.java:95-125): "a" + i % COUNT strings, and
    # testBinaryDictionaryFallBack-style longer entries; empty and 1-entry dictionaries
    for values in ([("a%d" % (i % 100)).encode() for i in range(100)], [b""], [], [b"x" * 70000, b"", b"yz"]):
        page = orc.binary_dict_page(values)
        st, v = orc.binary_dict_views(page, len(values))
        st2, v2 = lib_views(page, len(values))
        assert st == 0 and st2 == 0 and (v == v2).all()
        assert [page[int(x) & 0xFFFFFFFF:(int(x) & 0xFFFFFFFF) + (int(x) >> 32)].tobytes() for x in v] == values


def test_views_fixed_length_and_malformed_pages(orc):
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 256, 12 * 7, dtype=np.uint8)
    for fixed, n in ((12, 7), (12, 8), (5, 16), (1, 84), (84, 1), (85, 1)):
        st, v = orc.binary_dict_views(raw, n, fixed)
        st2, v2 = lib_views(raw, n, fixed)
        assert st == st2 == 0 and (v == v2).all()
        for i, x in enumerate(v):  # entries past the page only fail when they are read
            assert (int(x) == INVALID) == ((i + 1) * fixed > raw.size)
    assert lib_views(raw, 3, -1)[0] == 2 and orc.binary_dict_views(raw, 3, -1)[0] == 2
    page = orc.binary_dict_page([b"abc", b"defgh", b"ij"])
    # the last value cut short: its view is invalid, the page still parses (Binary objects are lazy)
    for cut in (1, 2):
        st, v = lib_views(page[:-cut], 3)
        st2, v2 = orc.binary_dict_views(page[:-cut], 3)
        assert st == st2 == 0 and (v == v2).all() and int(v[2]) == INVALID and int(v[1]) != INVALID
    # a length prefix outside the page: IndexOutOfBoundsException in the constructor
    for cut in (3, 4, 5, 6):
        assert lib_views(page[:-cut], 3)[0] == 4 and orc.binary_dict_views(page[:-cut], 3)[0] == 4
    assert lib_views(page, 4)[0] == 4 and orc.binary_dict_views(page, 4)[0] == 4
    # a negative length walks backwards, as Java's int arithmetic does
    neg = np.concatenate([orc.binary_dict_page([b"abcdefgh"]), np.frombuffer((-8 & 0xFFFFFFFF).to_bytes(4, "little"), np.uint8),
                          ]).astype(np.uint8)
    st, v = lib_views(neg, 3)
    st2, v2 = orc.binary_dict_views(neg, 3)
    assert st == st2 == 0 and (v == v2).all() and int(v[1]) == INVALID
    assert int(v[2]) == (0x64636261 << 32 | 12) or int(v[2]) == INVALID  # "abcd" read as a length: far outside
    rng = np.random.default_rng(6)
    for _ in range(200):  # random bytes: both walks agree on status and views
        junk = rng.integers(0, 256, int(rng.integers(0, 64)), dtype=np.uint8)
        junk[rng.integers(0, 2, junk.size).astype(bool)] = 0
        n = int(rng.integers(0, 8))
        st, v = lib_views(junk, n)
        st2, v2 = orc.binary_dict_views(junk, n)
        assert st == st2 and (st != 0 or (v == v2).all())


def test_oracle_materialize(orc):
    values = [b"abc", b"", b"hello world", b"x"]
    page = orc.binary_dict_page(values)
    st, views = orc.binary_dict_views(page, 4)
    ids = [0, 2, 2, 1, 3]
    st, off, out = orc.binary_materialize(views[ids], page, 100)
    assert st == 0 and off.tolist() == [0, 3, 14, 25, 25, 26] and out.tobytes() == b"abchello worldhello worldx"
    st, off, out = orc.binary_materialize(views[ids], page, 24)  # the third value does not fit
    assert st == 2 and off.tolist() == [0, 3, 14] and out.tobytes() == b"abchello world"
    bad = views[ids].copy()
    bad[3] = INVALID
    st, off, out = orc.binary_materialize(bad, page, 100)
    assert st == 4 and off.tolist() == [0, 3, 14, 25]


# ------------------------------------------------------------------------------- GPU
def check_pages(gh, orc, cases):
    """cases: (views, dict_bytes, capacity)"""
    res = gh.materialize_binary(cases)
    for (views, dict_bytes, cap), (st, written, nbytes, offsets, data, guards) in zip(cases, res):
        ost, ooff, oout = orc.binary_materialize(views, dict_bytes, cap)
        assert st == ost, (st, ost, len(views))
        assert written == len(ooff) - 1 and nbytes == oout.size
        assert (offsets[:written + 1] == ooff).all()
        assert (data[:nbytes] == oout).all()
        for g in guards:
            assert (g == 0xA5).all()  # nothing written outside offsets[0..n] and bytes[0..capacity)


@pytest.mark.gpu
def test_gpu_materialize_matches_oracle(orc):
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(11)
    cases = []
    for n_dict, max_len, n in ((50, 12, 20000), (1000, 40, 5000), (7, 3, 2049), (3, 0, 100), (20, 16, 2048),
                               (5, 70000, 40), (300, 200, 3000), (64, 9, 1), (64, 9, 0), (2, 17, 7)):
        values = random_dictionary(rng, n_dict, max_len)
        page = orc.binary_dict_page(values)
        st, views = orc.binary_dict_views(page, n_dict)
        ids = rng.integers(0, n_dict, n)
        ids[:min(n, n_dict)] = np.arange(n_dict)[:min(n, n_dict)][::-1]  # the last entries too (dictionary tail)
        total = int(sum(len(values[i]) for i in ids))
        cases.append((views[ids], page, total))          # exact fit
        cases.append((views[ids], page, total + 37))
        if total > 1:
            cases.append((views[ids], page, total // 2))  # PQD_BAD_ARG part way
            cases.append((views[ids], page, 0))
        if n > 3:
            bad = views[ids].copy()
            bad[n // 2] = INVALID
            bad[n - 1] = (np.uint64(5) << np.uint64(32)) | np.uint64(page.size - 2)  # runs past the dictionary
            cases.append((bad, page, total + 8))
            bad2 = views[ids].copy()
            bad2[n - 1] = (np.uint64(5) << np.uint64(32)) | np.uint64(page.size - 2)
            cases.append((bad2, page, total + 8))
    check_pages(gh, orc, cases)


@pytest.mark.gpu
def test_gpu_binary_dictionary_pages_end_to_end(orc, dict_path):
    """ids -> views (the dictionary gather with a dictionary of views) -> bytes, many pages per launch."""
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(12)
    values = random_dictionary(rng, 5000, 12)
    page = orc.binary_dict_page(values)
    st, views = orc.binary_dict_views(page, len(values))
    assert st == 0
    pages, ids_all = [], []
    for p in range(40):
        n = int(rng.integers(1, 30000))
        ids = rng.integers(0, len(values), n).astype(np.int32)
        if p % 3 == 0:
            ids[rng.integers(0, n):] = ids[0]  # a long RLE run at the end
        ids_all.append(ids)
        pages.append(gh.Page(orc.dict_encode_ids(ids, len(values)), n, pq.ENC_RLE_DICTIONARY, 8, dictionary=views))
    res = gh.decode_pages(pages)
    cases = []
    for ids, (st, written, _, out, _) in zip(ids_all, res):
        got = out.view(np.uint64)
        assert st == 0 and written == ids.size and (got == views[ids]).all()
        cases.append((got, page, int((got >> np.uint64(32)).sum())))
    check_pages(gh, orc, cases)
    for ids, (st, written, nbytes, offsets, data, _) in zip(ids_all[:5], gh.materialize_binary(cases[:5])):
        assert [data[offsets[i]:offsets[i + 1]].tobytes() for i in range(0, ids.size, 997)] == \
               [values[j] for j in ids[::997]]


@pytest.mark.gpu
def test_gpu_reader_mirror_reads_java_written_binary_pages(orc):
    """DictionaryValuesReader over a PlainBinaryDictionary built from the dictionary page parquet-java
    wrote: readBytes() yields the strings pyarrow reads from the same file."""
    import parquet_java_amd as pq
    pq.init(0)
    for f in java_dictionaries():
        d = pq.PlainBinaryDictionary(bytes.fromhex(f["dictionary_page"]), len(f["dictionary"]))
        assert d.getMaxId() == len(f["dictionary"]) - 1
        assert d.decodeToBinary(0) == f["dictionary"][0].encode("latin-1")
        r = pq.DictionaryValuesReader(d)
        n = len(f["expected"])
        r.initFromPage(n, pq.ByteBufferInputStream.wrap(bytes.fromhex(f["data"])))
        got = [r.readBytes() for _ in range(n)]
        assert got == [f["dictionary"][i].encode("latin-1") for i in f["expected"]], f["column"]
        with pytest.raises(pq.IllegalArgumentException):
            r.readBytes()  # "Reading past RLE/BitPacking stream."
    # an id beyond the dictionary: the values before it are delivered, then the read fails
    values = [b"a", b"bcd", b""]
    d = pq.PlainBinaryDictionary(orc.binary_dict_page(values).tobytes(), 3)
    page = orc.dict_encode_ids(np.array([0, 1, 2, 1, 3, 0], np.int32), 4)
    r = pq.DictionaryValuesReader(d)
    r.initFromPage(6, pq.ByteBufferInputStream.wrap(page.tobytes()))
    assert [r.readBytes() for _ in range(4)] == [b"a", b"bcd", b"", b"bcd"]
    with pytest.raises(pq.ParquetDecodingException):
        r.readBytes()
    # FIXED_LEN_BYTE_ARRAY dictionary (PlainBinaryDictionary(page, length))
    raw = np.arange(40, dtype=np.uint8)
    d = pq.PlainBinaryDictionary(raw.tobytes(), 8, 5)
    r = pq.DictionaryValuesReader(d)
    ids = np.array([7, 0, 3, 3, 3, 3, 3, 3, 3, 3, 3, 1], np.int32)
    r.initFromPage(ids.size, pq.ByteBufferInputStream.wrap(orc.dict_encode_ids(ids, 8).tobytes()))
    assert [r.readBytes() for _ in range(ids.size)] == [raw[i * 5:i * 5 + 5].tobytes() for i in ids]
    with pytest.raises(pq.IllegalArgumentException):
        pq.PlainBinaryDictionary(raw.tobytes(), 8, 0)


@pytest.mark.gpu
def test_gpu_materialize_garbage_views(orc):
    """Views that point anywhere: outside the dictionary, overlapping, empty, long, invalid."""
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(13)
    cases = []
    for k in range(120):
        dict_len = int(rng.choice([0, 1, 15, 16, 17, 100, 5000]))
        dict_bytes = rng.integers(0, 256, dict_len, dtype=np.uint8)
        n = int(rng.integers(0, 6000))
        off = rng.integers(0, dict_len + (12 if k % 4 == 0 else 1), n).astype(np.uint64)
        ln = rng.integers(0, int(rng.choice([1, 9, 17, 40, 3000])), n).astype(np.uint64)
        if k % 4:
            ln = np.minimum(ln, np.uint64(dict_len) - np.minimum(off, np.uint64(dict_len)))  # mostly valid views
        views = (ln << np.uint64(32)) | off
        if k % 7 == 0 and n:
            views[rng.integers(0, n, 2)] = INVALID
        total = int(ln.sum())
        cases.append((views, dict_bytes, int(rng.choice([total, total // 2, total + 5, 0]))))
    check_pages(gh, orc, cases)


# ------------------------------------------------------------------------------- fixed-width dictionaries
@pytest.mark.gpu
@pytest.mark.parametrize("type_name,dtype", [("TYPE_INT32", np.int32), ("TYPE_INT64", np.int64),
                                             ("TYPE_FLOAT", np.float32), ("TYPE_DOUBLE", np.float64)])
def test_gpu_fixed_width_dictionary_handles(orc, dict_path, type_name, dtype):
    """pqd_dict_create for INT32 / INT64 / FLOAT / DOUBLE (PlainValuesDictionary.PlainIntegerDictionary,
    PlainLongDictionary, PlainFloatDictionary, PlainDoubleDictionary: PlainValuesDictionary.java:139-346):
    the dictionary page body is dictionarySize little-endian values; pages decoded through the handle's
    device array equal the oracle's DictionaryValuesReader, NaN payloads and -0.0 included
    (TestDictionary.java:475-505)."""
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(21)
    elem = np.dtype(dtype).itemsize
    for n_dict in (1, 50, 4097, 65536):
        raw = rng.integers(0, 256, n_dict * elem, dtype=np.uint8)  # any bit pattern: NaNs of every payload among them
        dic = raw.view(dtype).copy()
        if np.issubdtype(dtype, np.floating) and n_dict >= 50:
            bits = dic.view(np.uint32 if elem == 4 else np.uint64)
            bits[:4] = [0x7FC00000, 0x7FC00001, 0xFFC00000, 0x80000000] if elem == 4 else \
                [0x7FF8000000000000, 0x7FF8000000000001, 0xFFF8000000000000, 0x8000000000000000]
        page = dic.view(np.uint8)
        handle = C.c_void_p()
        st = pq.lib.pqd_dict_create(getattr(pq, type_name), 0, page.ctypes.data_as(C.c_void_p), page.size, n_dict, C.byref(handle))
        assert st == 0
        info = pq.DictInfo()
        assert pq.lib.pqd_dict_get(handle, C.byref(info)) == 0
        assert info.n_entries == n_dict and info.elem_size == elem and not info.bytes and info.bytes_len == 0
        assert info.physical_type == getattr(pq, type_name)
        # the device array holds the page's values
        back = np.zeros(n_dict * elem, np.uint8)
        pq.check(pq.lib.pqd_memcpy_d2h(back.ctypes.data_as(C.c_void_p), info.entries, back.size, None))
        torch.cuda.synchronize()
        assert (back == page).all()
        # pages decoded against it (device-resident batch, the handle's array as `dict`)
        datas, ids_all = [], []
        for n in (1, 777, 20000):
            ids = rng.integers(0, n_dict, n).astype(np.int32)
            ids[: min(n, 4)] = np.arange(min(n, 4)) % n_dict
            ids_all.append(ids)
            datas.append(orc.dict_encode_ids(ids, n_dict))
        d_blob = gh.to_dev(np.concatenate(datas))
        d_out = torch.zeros(sum(i.size for i in ids_all) * elem, dtype=torch.uint8, device=gh.dev())
        descs, pos, opos = [], 0, 0
        for ids, d in zip(ids_all, datas):
            descs.append(dict(data=d_blob.data_ptr() + pos, out=d_out.data_ptr() + opos, dict=info.entries, data_len=d.size,
                              num_values=ids.size, dict_len=info.n_entries, encoding=pq.ENC_RLE_DICTIONARY, elem_size=elem))
            pos += d.size
            opos += ids.size * elem
        results = (pq.PageResult * len(descs))()
        assert pq.lib.pqd_decode_pages(pq.make_descs(descs), results, len(descs), None) == 0
        got = d_out.cpu().numpy()
        opos = 0
        for ids, d, r in zip(ids_all, datas, results):
            ost, ovals = orc.dict_decode(d, dic, ids.size)
            assert r.status == ost == 0 and r.values_written == ids.size
            assert (got[opos:opos + ids.size * elem] == ovals.view(np.uint8)).all()  # bit patterns, not float compares
            opos += ids.size * elem
        assert pq.lib.pqd_dict_destroy(handle) == 0
    # a page shorter than dictionarySize values: the Java constructor runs off the page
    # (PlainValuesReader.readInteger -> ParquetDecodingException, PlainValuesReader.java:62-68)
    short = np.zeros(10 * elem - 1, np.uint8)
    handle = C.c_void_p()
    assert pq.lib.pqd_dict_create(getattr(pq, type_name), 0, short.ctypes.data_as(C.c_void_p), short.size, 10, C.byref(handle)) == pq.DECODE
    assert not handle.value
    # an empty dictionary is legal; a page that refers to it fails at its first value
    assert pq.lib.pqd_dict_create(getattr(pq, type_name), 0, None, 0, 0, C.byref(handle)) == 0
    assert pq.lib.pqd_dict_get(handle, C.byref(info)) == 0 and info.n_entries == 0
    assert pq.lib.pqd_dict_destroy(handle) == 0
    assert pq.lib.pqd_dict_create(99, 0, short.ctypes.data_as(C.c_void_p), short.size, 1, C.byref(handle)) == pq.BAD_ARG


@pytest.mark.gpu
def test_gpu_binary_dictionary_of_64k_entries(orc, dict_path):
    """The BINARY column of the 64 Ki-entry configuration: entries of 4-12 bytes (a 768 KiB dictionary
    page), ids -> views -> offsets + bytes, against the oracle's materialisation."""
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(31)
    n_dict = 65536
    lens = rng.integers(4, 13, n_dict)
    blob = rng.integers(0, 256, int(lens.sum()), dtype=np.uint8)
    ends = np.cumsum(lens)
    values = [blob[e - l:e].tobytes() for e, l in zip(ends, lens)]
    page = orc.binary_dict_page(values)
    assert 700_000 < page.size < 1 << 20
    st, views = orc.binary_dict_views(page, n_dict)
    assert st == 0
    pages, ids_all = [], []
    for p in range(12):
        n = 20000 if p < 10 else int(rng.integers(1, 20000))
        ids = rng.integers(0, n_dict, n).astype(np.int32)
        if p == 3:
            ids = np.sort(ids)
        if p == 4:
            ids[:] = n_dict - 1  # the dictionary's last entry, as one RLE run
        ids_all.append(ids)
        pages.append(gh.Page(orc.dict_encode_ids(ids, n_dict), n, pq.ENC_RLE_DICTIONARY, 8, dictionary=views))
    res = gh.decode_pages(pages)
    cases = []
    for ids, (st, written, _, out, _) in zip(ids_all, res):
        got = out.view(np.uint64)
        assert st == 0 and written == ids.size and (got == views[ids]).all()
        cases.append((got, page, int(lens[ids].sum())))
    check_pages(gh, orc, cases)
    st, written, nbytes, offsets, data, _ = gh.materialize_binary(cases[:1])[0]
    ids = ids_all[0]
    assert all(data[offsets[i]:offsets[i + 1]].tobytes() == values[ids[i]] for i in range(0, ids.size, 501))
