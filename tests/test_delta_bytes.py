"""DELTA_LENGTH_BYTE_ARRAY and DELTA_BYTE_ARRAY value pages.  The oracle (oracle/pqd_oracle_pages.c restates
DeltaLengthByteArrayValuesReader.java:44-58 and DeltaByteArrayReader.java:50-95) is checked against column
chunks written by pyarrow (an independent writer of the same format; the values are what pyarrow reads
back) and against a per-value Python restatement; the HIP kernels are checked against the oracle on the
same chunks, on synthetic pages (long values, long prefixes, pages linked through the previous value) and
on damaged pages."""
import ctypes as C

import numpy as np
import pytest

import pagefile as pf

FIX = [fx for fx in pf.load_fixtures() if fx["file"].startswith("delta_strings")]
ENC = {"DELTA_LENGTH_BYTE_ARRAY": 8, "DELTA_BYTE_ARRAY": 9}


def value_sections(fx):
    """[(encoding, values section, non-null count)] of the chunk's data pages (V2, uncompressed)."""
    out = []
    for page in pf.walk_chunk(pf.fixture_chunk(fx)):
        assert page.type == pf.DATA_PAGE_V2
        r, d, body = pf.split_v2(page)
        out.append((page.encoding, np.frombuffer(body, np.uint8), page.num_values - page.num_nulls))
    return out


# ---- writers (DeltaLengthByteArrayValuesWriter.java:44-99, DeltaByteArrayWriter.java:38-100): lengths as one
# DELTA_BINARY_PACKED block of ints (128 per block, 4 miniblocks), then the bytes
def write_dlba(orc, values):
    lens = np.array([len(v) for v in values], np.int32)
    return np.concatenate([orc.delta_encode(lens), np.frombuffer(b"".join(values), np.uint8)])


def write_dba(orc, values, previous=b""):
    prefixes, suffixes = [], []
    for v in values:
        k = 0
        while k < min(len(v), len(previous)) and v[k] == previous[k]:
            k += 1
        prefixes.append(k)
        suffixes.append(v[k:])
        previous = v
    return np.concatenate([orc.delta_encode(np.array(prefixes, np.int32)), write_dlba(orc, suffixes)])


def split(offsets, data):
    data = bytes(data)
    return [data[a:b] for a, b in zip(offsets[:-1], offsets[1:])]


def python_delta_byte_array(orc, section, n, previous=b""):
    """Value by value, as DeltaByteArrayReader.readBytes does it."""
    st, prefixes, total, used = orc.delta_decode(section, n)
    assert st == 0
    st, suffix_lens, total2, used2 = orc.delta_decode(section[used:], n)
    assert st == 0
    pos, out = used + used2, []
    for p, s in zip(prefixes.tolist(), suffix_lens.tolist()):
        suffix = bytes(section[pos:pos + s])
        pos += s
        previous = previous[:p] + suffix if p else suffix
        out.append(previous)
    return out, pos


# ------------------------------------------------------------------------------- oracle
@pytest.mark.parametrize("fx", FIX, ids=lambda fx: fx["file"] + "." + fx["column"])
def test_oracle_reads_pyarrow_delta_strings(orc, fx):
    got = []
    for enc, section, n in value_sections(fx):
        if enc == "DELTA_LENGTH_BYTE_ARRAY":
            st, offs, data, used = orc.delta_length_byte_array(section, n)
        else:
            assert enc == "DELTA_BYTE_ARRAY"
            st, offs, data, used = orc.delta_byte_array(section, n, 1 << 22)
            assert python_delta_byte_array(orc, section, n)[0] == split(offs, data)
        assert st == 0 and used == section.size and offs.size == n + 1
        got += split(offs, data)
    assert got == pf.fixture_values(fx)


def test_oracle_delta_byte_array_known_answers(orc):
    # the example of the format description: "axis", "axle", "babble", "babyhood"
    values = [b"axis", b"axle", b"babble", b"babyhood"]
    page = write_dba(orc, values)
    st, prefixes, _, used = orc.delta_decode(page, 4)
    assert prefixes.tolist() == [0, 2, 0, 3]
    st, offs, data, used2 = orc.delta_byte_array(page, 4, 100)
    assert st == 0 and split(offs, data) == values and used2 == page.size
    # a page that goes on from the value before it (files written before PARQUET-246 was fixed)
    page = write_dba(orc, [b"babylon", b"bat"], previous=b"babyhood")
    assert orc.delta_decode(page, 2)[1].tolist() == [4, 2]
    st, offs, data, _ = orc.delta_byte_array(page, 2, 100, prev=b"babyhood")
    assert st == 0 and split(offs, data) == [b"babylon", b"bat"]
    st, offs, data, _ = orc.delta_byte_array(page, 2, 100)  # without it the first prefix has nothing to copy from
    assert st == orc.DECODE and offs.tolist() == [0]
    # capacity, short pages, short length blocks
    page = write_dlba(orc, [b"ab", b"", b"cde"])
    st, offs, data, used = orc.delta_length_byte_array(page, 3)
    assert st == 0 and offs.tolist() == [0, 2, 2, 5] and bytes(data) == b"abcde"
    st, offs, data, used = orc.delta_length_byte_array(page, 3, cap=4)
    assert st == orc.BAD_ARG and offs.tolist() == [0, 2, 2]
    st, offs, data, used = orc.delta_length_byte_array(page[:-1], 3)
    assert st == orc.DECODE and offs.tolist() == [0, 2, 2] and bytes(data) == b"ab"
    st, offs, data, used = orc.delta_length_byte_array(page, 5)  # the block holds three lengths
    assert st == orc.DECODE and offs.tolist() == [0, 2, 2, 5]
    st, offs, data, used = orc.delta_length_byte_array(page, 2)  # more lengths than values: as the value readers
    assert st == orc.DECODE and offs.tolist() == [0]


# ------------------------------------------------------------------------------- GPU
def run_pages(pq, gh, torch, pages, misalign=1):
    """pages: (encoding, section, num_values, capacity, previous or None); one pqd_delta_byte_array_dev call.
    `previous` is a list of values decoded before (given to the page as offsets + bytes)."""
    n = len(pages)
    descs = (pq.DeltaBytesDesc * n)()
    keep = []
    for i, (enc, section, num_values, cap, previous) in enumerate(pages):
        section = np.ascontiguousarray(section, np.uint8)
        d_data = gh.to_dev(np.concatenate([np.zeros(misalign, np.uint8), section]))
        d_offs = torch.full((num_values + 3,), -77, dtype=torch.int32, device=gh.dev())
        d_bytes = torch.full((cap + 32 + misalign,), 0xA5, dtype=torch.uint8, device=gh.dev())
        d_work = torch.zeros(pq._native.delta_bytes_work(num_values) + 16, dtype=torch.uint8, device=gh.dev())
        keep.append((d_data, d_offs, d_bytes, d_work))
        descs[i].data = d_data.data_ptr() + misalign
        descs[i].offsets = d_offs.data_ptr()
        descs[i].bytes = d_bytes.data_ptr() + misalign
        descs[i].work = d_work.data_ptr()
        descs[i].data_len = section.size
        descs[i].num_values = num_values
        descs[i].bytes_capacity = cap
        descs[i].encoding = enc
        if previous:
            po = np.concatenate([[0], np.cumsum([len(v) for v in previous])]).astype(np.int32)
            d_po, d_pb = gh.to_dev(po), gh.to_dev(np.frombuffer(b"".join(previous) + b"\0", np.uint8))
            keep.append((d_po, d_pb))
            descs[i].prev_offsets, descs[i].prev_bytes, descs[i].prev_count = d_po.data_ptr(), d_pb.data_ptr(), len(previous)
    d_descs = gh.to_dev(np.frombuffer(bytes(descs), np.uint8).copy())
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=gh.dev())
    assert pq.lib.pqd_delta_byte_array_dev(d_descs.data_ptr(), d_res.data_ptr(), n, None) == 0
    torch.cuda.synchronize()
    raw = d_res.cpu().numpy().view(np.uint32).reshape(-1, 4)
    out = []
    for i, (enc, section, num_values, cap, previous) in enumerate(pages):
        st, written, consumed, nbytes = (int(x) for x in raw[i])
        d_data, d_offs, d_bytes, d_work = keep[[k for k, t in enumerate(keep) if len(t) == 4][i]]
        offs = d_offs.cpu().numpy()
        data = d_bytes.cpu().numpy()
        assert (offs[written + 1:] == -77).all(), "offsets written behind the last value"
        assert (data[:misalign] == 0xA5).all() and (data[misalign + nbytes:] == 0xA5).all(), "bytes written outside the values"
        out.append((st, offs[:written + 1], data[misalign:misalign + nbytes], consumed))
    return out


def check_against_oracle(orc, pages, results):
    for (enc, section, num_values, cap, previous), (st, offs, data, consumed) in zip(pages, results):
        if enc == ENC["DELTA_LENGTH_BYTE_ARRAY"]:
            ost, ooffs, odata, oused = orc.delta_length_byte_array(section, num_values, cap)
        else:
            ost, ooffs, odata, oused = orc.delta_byte_array(section, num_values, cap, prev=previous[-1] if previous else b"")
        assert st == ost, (st, ost)
        assert offs.tolist() == ooffs.tolist()
        assert bytes(data) == bytes(odata)
        if ost == 0 or offs.size > 1:
            assert consumed == oused


@pytest.mark.gpu
@pytest.mark.parametrize("fx", FIX, ids=lambda fx: fx["file"] + "." + fx["column"])
def test_gpu_reads_pyarrow_delta_strings(orc, fx):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    sections = value_sections(fx)
    pages = [(ENC[enc], section, n, 1 << 20, None) for enc, section, n in sections]
    results = run_pages(pq, gh, torch, pages)
    check_against_oracle(orc, pages, results)
    got = []
    for (st, offs, data, consumed), (enc, section, n) in zip(results, sections):
        assert st == 0 and consumed == section.size
        got += split(offs, data)
    assert got == pf.fixture_values(fx)


def random_values(rng, n, kind):
    if kind == "short":
        return [bytes(rng.integers(97, 101, rng.integers(0, 12), dtype=np.uint8)) for _ in range(n)]
    if kind == "sorted":
        return sorted(bytes(rng.integers(97, 99, rng.integers(0, 40), dtype=np.uint8)) for _ in range(n))
    if kind == "urls":  # long shared prefixes
        base = b"https://example.org/" + bytes(rng.integers(97, 123, 60, dtype=np.uint8))
        return sorted(base[:rng.integers(20, 81)] + b"/%d" % rng.integers(0, 1000) for _ in range(n))
    if kind == "long":  # values and prefixes beyond the LDS tiles
        base = bytes(rng.integers(0, 256, 40000, dtype=np.uint8))
        out = []
        for _ in range(n):
            k = int(rng.choice([0, 3, 900, 1023, 1024, 1025, 5000, 8191, 8192, 8193, 20000, 40000]))
            out.append(base[:k] + bytes(rng.integers(0, 256, rng.integers(0, 6), dtype=np.uint8)))
        return out
    if kind == "mixed":
        out = []
        for _ in range(n):
            r = rng.random()
            if r < 0.02:
                out.append(bytes(rng.integers(0, 256, rng.integers(3000, 12000), dtype=np.uint8)))
            elif r < 0.5 and out:
                out.append(out[-1][:rng.integers(0, len(out[-1]) + 1)] + bytes(rng.integers(0, 256, rng.integers(0, 20), dtype=np.uint8)))
            else:
                out.append(bytes(rng.integers(0, 256, rng.integers(0, 200), dtype=np.uint8)))
        return out
    raise ValueError(kind)


@pytest.mark.gpu
def test_gpu_delta_byte_arrays_against_oracle(orc):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(97)
    pages = []
    for kind, n in (("short", 0), ("short", 1), ("short", 63), ("short", 64), ("short", 65), ("short", 5000), ("sorted", 20000),
                    ("urls", 3000), ("long", 40), ("mixed", 700), ("mixed", 64)):
        values = random_values(rng, n, kind)
        total = sum(len(v) for v in values)
        pages.append((ENC["DELTA_LENGTH_BYTE_ARRAY"], write_dlba(orc, values), n, total, None))
        pages.append((ENC["DELTA_BYTE_ARRAY"], write_dba(orc, values), n, total, None))
        if n > 2:
            pages.append((ENC["DELTA_BYTE_ARRAY"], write_dba(orc, values), n, total - 1, None))  # the last value does not fit
            pages.append((ENC["DELTA_LENGTH_BYTE_ARRAY"], write_dlba(orc, values), n, total // 2, None))
            pages.append((ENC["DELTA_BYTE_ARRAY"], write_dba(orc, values)[:-1], n, total, None))  # the page ends early
            pages.append((ENC["DELTA_LENGTH_BYTE_ARRAY"], write_dlba(orc, values), n + 3, total, None))  # lengths run out
            # the page goes on from the values of the page before it
            before = random_values(rng, 5, kind)
            pages.append((ENC["DELTA_BYTE_ARRAY"], write_dba(orc, values, previous=before[-1]), n, total, before))
    for misalign in (1, 16):
        results = run_pages(pq, gh, torch, pages, misalign)
        check_against_oracle(orc, pages, results)
    # all right pages really are: spot check against the values themselves
    values = random_values(rng, 1000, "urls")
    (st, offs, data, consumed), = run_pages(pq, gh, torch, [(ENC["DELTA_BYTE_ARRAY"], write_dba(orc, values), 1000, 1 << 20, None)])
    assert st == 0 and split(offs, data) == values


@pytest.mark.gpu
def test_gpu_damaged_delta_byte_array_pages(orc):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(98)
    pages = []
    for trial in range(120):
        values = random_values(rng, int(rng.integers(1, 300)), ("short", "sorted", "mixed")[trial % 3])
        enc = (ENC["DELTA_LENGTH_BYTE_ARRAY"], ENC["DELTA_BYTE_ARRAY"])[trial % 2]
        page = (write_dlba if trial % 2 == 0 else write_dba)(orc, values).copy()
        for _ in range(int(rng.integers(1, 4))):  # damage in the length blocks, mostly
            at = int(rng.integers(0, min(page.size, 80)))
            page[at] = rng.integers(0, 256)
        if trial % 5 == 0:
            page = page[:rng.integers(0, page.size)]
        pages.append((enc, page, len(values), sum(len(v) for v in values) + 100, None))
    results = run_pages(pq, gh, torch, pages)
    check_against_oracle(orc, pages, results)


@pytest.mark.gpu
def test_gpu_delta_byte_array_readers(orc):
    import parquet_java_amd as pq
    pq.init(0)
    rng = np.random.default_rng(99)
    values = random_values(rng, 500, "urls")
    r = pq.DeltaLengthByteArrayValuesReader()
    stream = pq.ByteBufferInputStream.wrap(bytes(write_dlba(orc, values)) + b"tail")
    r.initFromPage(500, stream)
    assert [r.readBytes() for _ in range(3)] == values[:3]
    r.skip(490)
    assert [r.readBytes() for _ in range(7)] == values[493:]
    assert stream.available() == 4
    with pytest.raises(pq.ParquetDecodingException):
        r.readBytes()
    first = pq.DeltaByteArrayReader()
    first.initFromPage(300, pq.ByteBufferInputStream.wrap(bytes(write_dba(orc, values[:300]))))
    assert [first.readBytes() for _ in range(300)] == values[:300]
    second = pq.DeltaByteArrayReader()
    second.setPreviousReader(first)
    second.initFromPage(200, pq.ByteBufferInputStream.wrap(bytes(write_dba(orc, values[300:], previous=values[299]))))
    assert [second.readBytes() for _ in range(200)] == values[300:]
    offs, data = second.offsets_and_bytes()
    assert split(offs, data) == values[300:]
    # values that need more room than the first guess of the host entry
    big = [b"x" * 3000] + [b"x" * 3000 + b"%d" % i for i in range(200)]
    r = pq.DeltaByteArrayReader()
    r.initFromPage(len(big), pq.ByteBufferInputStream.wrap(bytes(write_dba(orc, big))))
    assert [r.readBytes() for _ in range(len(big))] == big
    with pytest.raises(pq.EOFException):  # a length block that ends inside its header fails initFromPage
        pq.DeltaByteArrayReader().initFromPage(5, pq.ByteBufferInputStream.wrap(b"\x80\x01\x04"))
    with pytest.raises(pq.ParquetDecodingException):  # so does one that announces more lengths than the page has values
        pq.DeltaLengthByteArrayValuesReader().initFromPage(2, pq.ByteBufferInputStream.wrap(bytes(write_dlba(orc, values[:9]))))
