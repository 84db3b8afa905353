"""PLAIN pages of BYTE_ARRAY values.  The oracle (oracle/pqd_oracle_pages.c: orc_plain_byte_array restates
BinaryPlainValuesReader.java:36-74) is pinned by pages parquet-java itself wrote (the `name` and `text` columns of the
reference's test resources, tests/golden/java_pages.json: expected values from a pure-Python reading that was
compared with pyarrow's reading of the same files when the fixtures were made) and by PLAIN chunks pyarrow wrote;
the HIP kernel is checked against the oracle on the same pages, on synthetic pages (values that cross the
kernel's windows and segments in every way, long values, empty values) and on damaged pages."""
import json
import os

import numpy as np
import pytest

import pagefile as pf
from test_delta_bytes import random_values, run_pages, split

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "java_pages.json")) as _f:
    JAVA = [f for f in json.load(_f)["fixtures"] if f["kind"] == "plain_byte_array"]
ARROW = [fx for fx in pf.load_fixtures() if fx["file"].startswith("plain_v") and fx["physical_type"] == "BYTE_ARRAY"]
ENC_PLAIN_BYTE_ARRAY = 10


def value_sections(orc, fx):
    """[(values section, non-null count)] of the data pages of an uncompressed pyarrow chunk."""
    out = []
    for page in pf.walk_chunk(pf.fixture_chunk(fx)):
        assert page.encoding == "PLAIN"
        if page.type == pf.DATA_PAGE_V2:
            r, d, body = pf.split_v2(page)
            out.append((np.frombuffer(body, np.uint8), page.num_values - page.num_nulls))
        else:
            r, d, body = pf.split_v1(page.body, fx["max_rep"], fx["max_def"])
            st, levels, _ = orc.rle_decode(pf.level_width(fx["max_def"]), d, page.num_values, prefixed=True)
            assert st == 0
            out.append((np.frombuffer(body, np.uint8), int((levels == fx["max_def"]).sum())))
    return out


def write_plain(values):
    return np.frombuffer(b"".join(len(v).to_bytes(4, "little") + v for v in values), np.uint8)


# ------------------------------------------------------------------------------- oracle
def test_fixtures_are_there():
    assert len(JAVA) >= 10 and {f["column"] for f in JAVA} == {"name", "text"}
    assert len(ARROW) == 4


@pytest.mark.parametrize("f", JAVA, ids=lambda f: "%s.%s.%d.%d" % (f["file"], f["column"], f["row_group"], f["page"]))
def test_oracle_reads_java_written_pages(orc, f):
    data = np.frombuffer(bytes.fromhex(f["data"]), np.uint8)
    want = [v.encode("latin-1") for v in f["expected"]]
    st, offs, out, used = orc.plain_byte_array(data, f["num_values"])
    assert st == 0 and used == data.size and split(offs, out) == want
    # the values stay where they are
    st, offs2, _, used = orc.plain_byte_array(data, f["num_values"], in_place=True)
    assert st == 0 and offs2.tolist() == offs.tolist()
    assert [bytes(data[o + 4 * (i + 1):o + 4 * (i + 1) + len(v)]) for i, (o, v) in enumerate(zip(offs2[:-1], want))] == want
    # one value more than the page holds: "could not read bytes at offset"
    st, offs, out, used = orc.plain_byte_array(data, f["num_values"] + 1)
    assert st == orc.DECODE and split(offs, out) == want and used == data.size


@pytest.mark.parametrize("fx", ARROW, ids=lambda fx: fx["file"] + "." + fx["column"])
def test_oracle_reads_pyarrow_plain_strings(orc, fx):
    got = []
    for section, n in value_sections(orc, fx):
        st, offs, out, used = orc.plain_byte_array(section, n)
        assert st == 0 and used == section.size and offs.size == n + 1
        got += split(offs, out)
    assert got == pf.fixture_values(fx)


def test_oracle_plain_byte_array_known_answers(orc):
    page = write_plain([b"ab", b"", b"cde"])
    assert bytes(page) == b"\x02\0\0\0ab\0\0\0\0\x03\0\0\0cde"
    st, offs, out, used = orc.plain_byte_array(page, 3)
    assert st == 0 and offs.tolist() == [0, 2, 2, 5] and bytes(out) == b"abcde" and used == 17
    st, offs, out, used = orc.plain_byte_array(page, 2)  # the page's last value is not asked for
    assert st == 0 and offs.tolist() == [0, 2, 2] and used == 10
    st, offs, out, used = orc.plain_byte_array(page, 3, cap=4)  # no room for the third
    assert st == orc.BAD_ARG and offs.tolist() == [0, 2, 2] and used == 10
    st, offs, out, used = orc.plain_byte_array(page[:-1], 3)  # the last value's bytes are cut
    assert st == orc.DECODE and offs.tolist() == [0, 2, 2] and bytes(out) == b"ab" and used == 10
    st, offs, out, used = orc.plain_byte_array(page[:12], 3)  # its length field is
    assert st == orc.DECODE and offs.tolist() == [0, 2, 2] and used == 10
    bad = page.copy()
    bad[9] = 0x80  # a negative length
    st, offs, out, used = orc.plain_byte_array(bad, 3)
    assert st == orc.DECODE and offs.tolist() == [0, 2] and used == 6
    st, offs, out, used = orc.plain_byte_array(page[:0], 0)
    assert st == 0 and offs.tolist() == [0] and used == 0
    st, offs, out, used = orc.plain_byte_array(page[:0], 1)
    assert st == orc.DECODE and offs.tolist() == [0]


# ------------------------------------------------------------------------------- GPU
def check_against_oracle(orc, pages, results):
    for k, ((enc, section, num_values, cap, _), (st, offs, data, consumed)) in enumerate(zip(pages, results)):
        ost, ooffs, odata, oused = orc.plain_byte_array(section, num_values, cap)
        assert st == ost, (k, st, ost)
        assert offs.tolist() == ooffs.tolist(), k
        assert bytes(data) == bytes(odata), k
        assert consumed == oused, k


@pytest.mark.gpu
def test_gpu_reads_java_written_and_pyarrow_pages(orc):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    pages, want = [], []
    for f in JAVA:
        data = np.frombuffer(bytes.fromhex(f["data"]), np.uint8)
        pages.append((ENC_PLAIN_BYTE_ARRAY, data, f["num_values"], data.size, None))
        want.append([v.encode("latin-1") for v in f["expected"]])
    for fx in ARROW:
        values, at = pf.fixture_values(fx), 0
        for section, n in value_sections(orc, fx):
            pages.append((ENC_PLAIN_BYTE_ARRAY, section, n, section.size, None))
            want.append(values[at:at + n])
            at += n
    results = run_pages(pq, gh, torch, pages)
    check_against_oracle(orc, pages, results)
    for (st, offs, data, consumed), (_, section, _, _, _), w in zip(results, pages, want):
        assert st == 0 and consumed == section.size and split(offs, data) == w


def shaped_values(rng, kind, n):
    if kind == "fixed":  # every value the same length: starts walk through all places of a segment
        k = int(rng.integers(0, 70))
        return [bytes(rng.integers(0, 256, k, dtype=np.uint8)) for _ in range(n)]
    if kind == "window_edges":  # length fields and values that straddle the 4096-byte windows in every way
        out, pos = [], 0
        while len(out) < n:
            to_edge = 4096 - pos % 4096
            k = int(rng.choice([0, 1, max(to_edge - 4, 0), max(to_edge - 3, 0), max(to_edge - 2, 0), max(to_edge - 1, 0), to_edge, to_edge + 1,
                                max(to_edge - 7, 0), max(to_edge - 5, 0), int(rng.integers(0, 200))]))
            out.append(bytes(rng.integers(0, 256, k, dtype=np.uint8)))
            pos += 4 + k
        return out
    if kind == "huge":  # values of many windows
        return [bytes(rng.integers(0, 256, int(rng.choice([0, 5, 4092, 4096, 8188, 8192, 12289, 30000, 70001])), dtype=np.uint8)) for _ in range(n)]
    return random_values(rng, n, kind)


@pytest.mark.gpu
def test_gpu_plain_byte_arrays_against_oracle(orc):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(131)
    pages = []
    for kind, n in (("short", 0), ("short", 1), ("short", 2), ("short", 63), ("short", 700), ("short", 20000), ("fixed", 1500), ("fixed", 1501),
                    ("fixed", 1502), ("fixed", 3000), ("window_edges", 400), ("window_edges", 401), ("huge", 30), ("mixed", 700), ("urls", 3000),
                    ("long", 40)):
        values = shaped_values(rng, kind, n)
        page = write_plain(values)
        total = sum(len(v) for v in values)
        pages.append((ENC_PLAIN_BYTE_ARRAY, page, n, total, None))
        pages.append((ENC_PLAIN_BYTE_ARRAY, np.concatenate([page, np.frombuffer(b"\x01\0\0\0x\xff\xff", np.uint8)]), n, total + 8, None))  # bytes behind the last value asked for
        if n > 2:
            pages.append((ENC_PLAIN_BYTE_ARRAY, page, n, total - 1, None))    # the last value does not fit
            pages.append((ENC_PLAIN_BYTE_ARRAY, page, n, total // 2, None))   # half of them do
            pages.append((ENC_PLAIN_BYTE_ARRAY, page, n, 0, None))
            pages.append((ENC_PLAIN_BYTE_ARRAY, page[:-1], n, total, None))   # the page ends early
            pages.append((ENC_PLAIN_BYTE_ARRAY, page[:page.size // 2], n, total, None))
            pages.append((ENC_PLAIN_BYTE_ARRAY, page, n + 3, total, None))    # more values asked for than there are
            pages.append((ENC_PLAIN_BYTE_ARRAY, page, n // 2, total, None))   # fewer
    for misalign in (1, 16):
        results = run_pages(pq, gh, torch, pages, misalign)
        check_against_oracle(orc, pages, results)
    values = shaped_values(rng, "window_edges", 2000)
    (st, offs, data, consumed), = run_pages(pq, gh, torch, [(ENC_PLAIN_BYTE_ARRAY, write_plain(values), 2000, sum(len(v) for v in values), None)])
    assert st == 0 and split(offs, data) == values


@pytest.mark.gpu
def test_gpu_damaged_plain_byte_array_pages(orc):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(132)
    pages = []
    for trial in range(150):
        values = shaped_values(rng, ("short", "fixed", "mixed", "window_edges")[trial % 4], int(rng.integers(1, 600)))
        page = write_plain(values).copy()
        for _ in range(int(rng.integers(1, 4))):
            page[int(rng.integers(0, page.size))] = rng.integers(0, 256)
        if trial % 5 == 0:
            page = page[:rng.integers(0, page.size)]
        pages.append((ENC_PLAIN_BYTE_ARRAY, page, len(values), sum(len(v) for v in values) + 100, None))
    pages += [(ENC_PLAIN_BYTE_ARRAY, rng.integers(0, 256, int(rng.integers(0, 9000)), dtype=np.uint8), 50, 10000, None) for _ in range(50)]
    pages += [(ENC_PLAIN_BYTE_ARRAY, rng.integers(0, 3, int(rng.integers(0, 9000)), dtype=np.uint8), 5000, 10000, None) for _ in range(50)]
    results = run_pages(pq, gh, torch, pages)
    check_against_oracle(orc, pages, results)


@pytest.mark.gpu
def test_gpu_plain_byte_arrays_left_in_place_and_mixed_calls(orc):
    """bytes == NULL: offsets only (the values are read where they lie); one call with pages of all three encodings."""
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    from test_delta_bytes import write_dba, write_dlba
    pq.init(0)
    rng = np.random.default_rng(133)
    values = shaped_values(rng, "mixed", 900)
    page = write_plain(values)
    d_page = gh.to_dev(page)
    d_offs = torch.full((903,), -77, dtype=torch.int32, device=gh.dev())
    desc = (pq.DeltaBytesDesc * 1)()
    desc[0].data, desc[0].offsets, desc[0].data_len, desc[0].num_values, desc[0].encoding = d_page.data_ptr(), d_offs.data_ptr(), page.size, 900, 10
    d_desc = gh.to_dev(np.frombuffer(bytes(desc), np.uint8).copy())
    d_res = torch.zeros(16, dtype=torch.uint8, device=gh.dev())
    assert pq.lib.pqd_delta_byte_array_dev(d_desc.data_ptr(), d_res.data_ptr(), 1, None) == 0
    torch.cuda.synchronize()
    st, written, consumed, nbytes = (int(x) for x in d_res.cpu().numpy().view(np.uint32))
    offs = d_offs.cpu().numpy()
    assert (st, written, consumed, nbytes) == (0, 900, page.size, sum(len(v) for v in values))
    assert [bytes(page[o + 4 * (i + 1):o + 4 * (i + 1) + len(v)]) for i, (o, v) in enumerate(zip(offs[:900], values))] == values
    assert offs[900] == nbytes and (offs[901:] == -77).all()
    total = sum(len(v) for v in values)
    pages = [(10, page, 900, total, None), (8, write_dlba(orc, values), 900, total, None), (9, write_dba(orc, values), 900, total, None),
             (10, page, 900, total, None)]
    for st, offs, data, consumed in run_pages(pq, gh, torch, pages):
        assert st == 0 and split(offs, data) == values


@pytest.mark.gpu
def test_gpu_binary_plain_values_reader(orc):
    import parquet_java_amd as pq
    pq.init(0)
    rng = np.random.default_rng(134)
    values = random_values(rng, 500, "urls")
    r = pq.BinaryPlainValuesReader()
    stream = pq.ByteBufferInputStream.wrap(bytes(write_plain(values)))
    r.initFromPage(500, stream)
    assert [r.readBytes() for _ in range(3)] == values[:3]
    r.skip(490)
    assert [r.readBytes() for _ in range(7)] == values[493:]
    with pytest.raises(pq.ParquetDecodingException):
        r.readBytes()
    offs, data = r.offsets_and_bytes()
    assert split(offs, data) == values
    r = pq.BinaryPlainValuesReader()  # a page that ends inside its third value: the first two are read, the third fails
    r.initFromPage(3, pq.ByteBufferInputStream.wrap(bytes(write_plain([b"ab", b"", b"cde"]))[:-1]))
    assert [r.readBytes(), r.readBytes()] == [b"ab", b""]
    with pytest.raises(pq.ParquetDecodingException):
        r.readBytes()
    big = [b"x" * 100000, b"", b"y" * 70000]  # more room than the page is long is never needed, but a first guess may be too small
    r = pq.BinaryPlainValuesReader()
    r.initFromPage(3, pq.ByteBufferInputStream.wrap(bytes(write_plain(big))))
    assert [r.readBytes() for _ in range(3)] == big
