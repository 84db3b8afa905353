"""Parity of the HIP decode path (called through the C ABI) with the CPU oracle, bit for bit.

All tests here need a real MI355X: run with ``-m gpu``.
"""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    import parquet_java_amd as pq
    assert pq.device_count() >= 1, "libpqdecode_hip.so sees no gfx950 device"
    pq.init(0)
    import gpu_harness
    return gpu_harness


def rand_vals(rng, w, n):
    if w == 0:
        return np.zeros(n, np.int32)
    return rng.integers(0, 1 << w, n, dtype=np.uint64).astype(np.uint32).view(np.int32)


# ------------------------------------------------------------------------------- bulk unpack
@pytest.mark.parametrize("w", range(0, 33))
def test_unpack_i32_all_widths(gpu, orc, w):
    rng = np.random.default_rng(42 + w)
    for n, off in ((0, 0), (8, 1), (64, 2), (1000, 3), (4096, 0), (100_000, 1)):
        n = n // 8 * 8
        vals = rand_vals(rng, w, n)
        packed = orc.pack(w, vals) if w else np.zeros(0, np.uint8)
        st, out, guard = gpu.unpack_dev(w, packed, n, offset=off)
        assert st == 0
        ost, oout, _ = orc.read_batch(w, packed, n)
        assert ost == 0 and (out == oout).all() and (out == vals).all(), (w, n)
        assert (guard == -0x5A5A5A5B).all()
    assert gpu.unpack_dev(w, np.zeros(w, np.uint8), 12)[0] == 2  # not a multiple of 8
    if w:
        assert gpu.unpack_dev(w, np.zeros(w * 2 - 1, np.uint8), 16)[0] == 1  # EOFException


@pytest.mark.parametrize("w", list(range(0, 65)))
def test_unpack_i64_all_widths(gpu, orc, w):
    rng = np.random.default_rng(1000 + w)
    for n, off in ((8, 0), (72, 3), (10_000, 1)):
        n = n // 8 * 8
        vals = (rng.integers(0, 1 << 63, n, dtype=np.uint64) * 2 + rng.integers(0, 2, n, dtype=np.uint64))
        vals = (vals & np.uint64((1 << w) - 1 if w < 64 else 0xFFFFFFFFFFFFFFFF)).view(np.int64) if w else np.zeros(n, np.int64)
        packed = orc.pack(w, vals, long=True) if w else np.zeros(0, np.uint8)
        st, out, guard = gpu.unpack_dev(w, packed, n, long=True, offset=off)
        ost, oout, _ = orc.read_batch(w, packed, n, long=True)
        assert st == 0 and ost == 0 and (out == oout).all() and (out == vals).all(), (w, n)
    assert gpu.unpack_dev(65, np.zeros(80, np.uint8), 8, long=True)[0] == 2


def test_unpack_full_size_round_trip(gpu, orc):
    # BASELINE config 1 size: 1 M values; size-independent property: pack(unpack(x)) == x on a sample + exact compare
    rng = np.random.default_rng(7)
    for w in (1, 7, 16, 23, 32):
        n = 1_000_000
        vals = rand_vals(rng, w, n)
        # numpy bit packing (LSB first) as an independent packer for the large case
        bits = ((vals.view(np.uint32)[:, None] >> np.arange(w, dtype=np.uint32)) & 1).astype(np.uint8)
        packed = np.packbits(bits.reshape(-1), bitorder="little")
        st, out, _ = gpu.unpack_dev(w, packed, n, offset=1)
        assert st == 0 and (out == vals).all()


# ------------------------------------------------------------------------------- RLE hybrid
def check_rle(gpu, orc, cases, elem=4):
    """cases: list of (w, data bytes, n, flags); compares every page with the oracle."""
    import parquet_java_amd as pq
    pages = [gpu.Page(d, n, pq.ENC_RLE, 4, bit_width=w, flags=f) for (w, d, n, f) in cases]
    res = gpu.decode_pages(pages)
    for (w, d, n, f), (st, written, consumed, out, guard) in zip(cases, res):
        if f & pq.F_WIDTH_PREFIX:
            ost, ovals = orc.dict_decode_ids(d, n)
            oused = None
        else:
            ost, ovals, oused = orc.rle_decode(w, d, n, prefixed=bool(f & pq.F_LENGTH_PREFIX))
        assert st == ost, (w, n, f, st, ost)
        assert written == ovals.size, (w, n, f, written, ovals.size)
        assert (out.view(np.int32)[:written] == ovals).all(), (w, n, f)
        if ost == 0 and oused is not None and n > 0:
            assert consumed == oused, (w, n, consumed, oused)
        assert (guard == 0xA5).all(), "wrote past the output"


def test_rle_integration_pattern_all_widths(gpu, orc, rle_path):
    from test_oracle_kat import integration_pattern
    cases = []
    for w in range(0, 33):
        vals = integration_pattern(w)
        cases.append((w, orc.rle_encode(w, vals), vals.size, 0))
        cases.append((w, orc.rle_encode(w, vals), vals.size - 5, 0))
    check_rle(gpu, orc, cases)


def test_rle_known_answer_and_quirks(gpu, orc, rle_path):
    import parquet_java_amd as pq
    kat = bytes([(1 << 1) | 1, (1 << 0) | (2 << 2) | (3 << 4)])
    cases = [
        (2, kat, 3, 0), (2, kat, 8, 0), (2, kat, 9, 0),  # truncated final run; read past the end
        (3, bytes([10 << 1, 0xFF]), 10, 0),  # RLE value not masked
        (8, bytes([0, 7, 4, 9]), 5, 0),  # zero-count RLE run repeats forever
        (8, bytes([1, 7]), 2, 0),  # zero-group packed run
        (16, bytes([4, 1]), 2, 0),  # truncated RLE value
        (1, bytes([0x90, 0x03, 1]), 200, 0),  # two-byte varint
        (33, b"\x02\x00", 1, 0),  # width > 32
        (5, b"", 0, 0), (5, b"", 3, 0),  # empty stream
        (0, bytes([20]), 10, 0), (0, bytes([5]), 16, 0), (0, bytes([5]), 17, 0),  # width 0
        (7, bytes([0x80, 0x80, 0x80]), 4, 0),  # varint runs off the end
        (4, bytes([3, 0x21]), 8, 0),  # packed run cut after one byte of four
        # headers longer than a 32-bit varint needs: BytesUtils.readUnsignedVarInt (BytesUtils.java:202-211)
        # reads on until a byte without the continuation bit, its shifts wrapping at 32
        (8, bytes([0x94, 0x80, 0x80, 0x80, 0x00, 9]), 10, 0),  # five bytes, still canonical: count 10
        (8, bytes([0x94] + [0x80] * 5 + [0x00, 9]), 10, 0),  # seven bytes, same value
        (8, bytes([0x80] * 5 + [0x05, 9]), 40, 0),  # byte 6 lands at shift 35 & 31 = 3: header 40, count 20
        (8, bytes([0x94] + [0x80] * 10 + [0x00, 9, 4, 3]), 12, 0),  # twelve bytes, then a second run
        (3, bytes([0x83] + [0x80] * 11 + [0x00, 0x88, 0xC6, 0xFA]), 8, 0),  # a bit-packed run behind a 13-byte header
        (8, bytes([0x94] + [0x80] * 300 + [0x01, 9]), 15, 0),  # 302 bytes: shift 7 * 301 & 31 = 27: count 10 + (1 << 26)
        (8, bytes([2, 5]) + bytes([0x90] + [0x80] * 40 + [0x00, 6]), 9, 0),  # a normal run first, then an over-long header
        (8, bytes([0x94] + [0x80] * 20), 4, 0),  # over-long and runs off the end
    ]
    body = orc.rle_encode(2, [1, 2, 3, 0] * 10)
    pre = struct.pack("<i", body.size) + bytes(body) + b"\xAA\xBB"
    cases += [(2, pre, 40, pq.F_LENGTH_PREFIX), (2, pre[:8], 40, pq.F_LENGTH_PREFIX), (2, pre[:3], 4, pq.F_LENGTH_PREFIX),
              (2, struct.pack("<i", -5) + b"\x00" * 8, 4, pq.F_LENGTH_PREFIX)]
    ids = np.array([i % 50 for i in range(2000)], np.int32)
    page = orc.dict_encode_ids(ids, 50)
    cases += [(0, page, 2000, pq.F_WIDTH_PREFIX), (0, b"", 0, pq.F_WIDTH_PREFIX), (0, b"", 5, pq.F_WIDTH_PREFIX)]
    check_rle(gpu, orc, cases)


def test_rle_random_mixed_runs(gpu, orc, rle_path):
    rng = np.random.default_rng(5)
    cases = []
    for w in (1, 2, 3, 5, 8, 9, 12, 16, 17, 24, 31, 32):
        for n in (1, 7, 8, 9, 503, 504, 505, 4097, 20_000):
            # run lengths ~ geometric(mean 12): mixes RLE and packed runs and odd boundaries
            vals = []
            while len(vals) < n:
                run = int(rng.geometric(1 / 12))
                v = int(rng.integers(0, 1 << w))
                vals += [v] * run if rng.random() < 0.5 else list(rng.integers(0, 1 << w, run))
            vals = np.array(vals[:n], np.uint64).astype(np.uint32).view(np.int32)
            enc = orc.rle_encode(w, vals)
            cases.append((w, enc, n, 0))
            cases.append((w, enc, n + 20, 0))  # reads past the encoded values: padding then PAST_END
            cases.append((w, enc[:enc.size * 2 // 3], n, 0))  # truncated stream
    check_rle(gpu, orc, cases)


def test_rle_large_pages_cross_windows(gpu, orc, rle_path):
    rng = np.random.default_rng(9)
    cases = []
    for w, n in ((32, 60_000), (20, 300_000), (1, 1_000_000), (13, 1_000_000)):
        vals = rand_vals(rng, w, n)
        vals[1000:5000] = 3 if w > 1 else 1  # a long RLE run in the middle
        cases.append((w, orc.rle_encode(w, vals), n, 0))
    # one huge bit-packed run (multi-byte header) and one huge RLE run
    w, n = 11, 200_000
    vals = rand_vals(rng, w, n)
    hdr = []
    h = ((n // 8) << 1) | 1
    while h >= 0x80:
        hdr.append((h & 0x7F) | 0x80)
        h >>= 7
    hdr.append(h)
    cases.append((w, np.concatenate([np.array(hdr, np.uint8), orc.pack(w, vals)]), n, 0))
    cases.append((w, np.concatenate([np.array(hdr, np.uint8), orc.pack(w, vals)[:70_000]]), n, 0))  # cut short: zeros
    cases.append((9, bytes([0x80, 0x9A, 0x0C, 0x05, 0x01]), 100_000, 0))
    check_rle(gpu, orc, cases)


def test_rle_many_tiny_runs(gpu, orc, rle_path):
    # more runs than one segment table holds: count-1 RLE runs and 1-group packed runs
    data = bytearray()
    for i in range(3000):
        data += bytes([2, i & 0xFF]) if i % 2 else bytes([3, i & 0xFF])
    check_rle(gpu, orc, [(8, bytes(data), 1500 + 1500 * 8, 0), (8, bytes(data), 7001, 0)])


def test_rle_many_pages_all_widths(gpu, orc, rle_path):
    """Hundreds of hybrid pages in one launch: every width 0..32, run boundaries off multiples of
    four, pages from one value to several chunks of the plan, length-prefixed and width-prefixed
    streams, streams cut short."""
    import parquet_java_amd as pq
    rng = np.random.default_rng(99)
    pages, expect = [], []
    for i in range(500):
        w = int(rng.integers(0, 33))
        n = int(rng.choice([1, 5, 8, 255, 256, 257, 4095, 4096, 20000, 32768, 32769, 70001])) if i % 2 else int(rng.integers(1, 50000))
        vals = np.empty(n, np.int64)
        pos = 0
        mean_run = int(rng.choice([1, 3, 9, 40, 600]))
        while pos < n:
            ln = min(n - pos, int(rng.geometric(1.0 / mean_run)))
            vals[pos:pos + ln] = rng.integers(0, 1 << w) if rng.random() < 0.5 else rng.integers(0, 1 << w, ln)
            pos += ln
        body = bytes(orc.rle_encode(w, vals.astype(np.uint32).view(np.int32)))
        kind = i % 5
        if i % 31 == 0 and len(body) > 2:
            body = body[:len(body) - int(rng.integers(1, min(len(body), 30)))]  # stream cut short
        if kind == 0:
            data, flags, bw = len(body).to_bytes(4, "little") + body + b"\x99\x98", pq.F_LENGTH_PREFIX, w
            ost, ovals, oused = orc.rle_decode(w, data, n, prefixed=True)
        elif kind == 1:
            data, flags, bw = bytes([w]) + body, pq.F_WIDTH_PREFIX, 0
            ost, ovals = orc.dict_decode_ids(data, n)
        else:
            data, flags, bw = body, 0, w
            ost, ovals, oused = orc.rle_decode(w, body, n)
        pages.append(gpu.Page(data, n, pq.ENC_RLE, 4, bit_width=bw, flags=flags))
        expect.append((ost, ovals, kind))
    res = gpu.decode_pages(pages)
    for i, ((st, written, consumed, out, guard), (ost, ovals, kind)) in enumerate(zip(res, expect)):
        assert st == ost and written == ovals.size, (i, kind, st, ost, written, ovals.size, pages[i].num_values)
        assert (out.view(np.int32)[:written] == ovals).all(), (i, kind)
        assert (guard == 0xA5).all(), i


# ------------------------------------------------------------------------------- dictionary
@pytest.mark.parametrize("dtype", [np.int32, np.int64, np.float32, np.float64])
def test_dictionary_pages(gpu, orc, dtype, dict_path):
    import parquet_java_amd as pq
    rng = np.random.default_rng(11)
    elem = np.dtype(dtype).itemsize
    pages, expect = [], []
    for dict_n, n in ((1, 100), (2, 1000), (50, 2000), (1000, 20_000), (65536, 20_000), (65536, 100_003), (300, 9)):
        raw = rng.integers(0, 1 << 63, dict_n, dtype=np.uint64)
        dic = raw.astype(np.uint32).view(dtype) if elem == 4 else raw.view(dtype)
        if dtype in (np.float32, np.float64) and dict_n >= 6:  # NaN payloads, -0.0, infinities survive
            special = [0x7FC00000, 0x7FC00001, 0xFFC00000, 0x7F800001, 0x80000000, 0x7F800000] if elem == 4 else \
                [0x7FF8000000000000, 0x7FF8000000000001, 0xFFF8000000000000, 0x7FF0000000000001, 1 << 63, 0x7FF0000000000000]
            dic.view(np.uint32 if elem == 4 else np.uint64)[:6] = special
        kind = rng.integers(0, 3)
        if kind == 0:
            ids = rng.integers(0, dict_n, n)
        elif kind == 1:
            ids = np.sort(rng.integers(0, dict_n, n))  # RLE-heavy
        else:
            ids = np.minimum((rng.zipf(1.1, n) - 1), dict_n - 1)
        ids = ids.astype(np.int32)
        page = orc.dict_encode_ids(ids, dict_n)
        pages.append(gpu.Page(page, n, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic))
        expect.append(orc.dict_decode(page, dic, n))
    # id beyond the dictionary, empty page, width byte > 32
    small = np.arange(5, dtype=np.uint64).astype(np.uint32).view(dtype) if elem == 4 else np.arange(5, dtype=np.uint64).view(dtype)
    bad = orc.dict_encode_ids(np.array([0, 1, 2, 9, 1] * 10, np.int32), 10)
    for page, n in ((bad, 50), (b"", 0), (b"", 4), (bytes([40, 2, 0]), 1)):
        pages.append(gpu.Page(page, n, pq.ENC_RLE_DICTIONARY, elem, dictionary=small))
        expect.append(orc.dict_decode(page, small, n))
    res = gpu.decode_pages(pages)
    ut = np.uint32 if elem == 4 else np.uint64
    for i, ((st, written, consumed, out, guard), (ost, ovals)) in enumerate(zip(res, expect)):
        assert st == ost and written == ovals.size, (i, st, ost, written, ovals.size)
        assert (out.view(ut)[:written] == ovals.view(ut)).all(), i
        assert (guard == 0xA5).all()


@pytest.mark.parametrize("elem", [4, 8])
def test_dictionary_pages_many_mixed(gpu, orc, elem, dict_path):
    """Hundreds of pages in one launch (each workgroup of the planned path walks several), with run
    boundaries off multiples of four, several dictionaries, truncated streams, ids beyond the
    dictionary, and pages the planned path hands to the generic kernel (too long, too many runs)."""
    import parquet_java_amd as pq
    rng = np.random.default_rng(1234 + elem)
    ut = np.uint32 if elem == 4 else np.uint64
    dicts = []
    for dn in (1, 3, 17, 300, 5000, 16384, 16385, 40000, 65536):
        raw = rng.integers(0, 1 << 63, dn, dtype=np.uint64)
        dicts.append(raw.astype(np.uint32) if elem == 4 else raw)
    pages, expect = [], []
    for i in range(700):
        dic = dicts[int(rng.integers(0, len(dicts)))]
        dn = dic.size
        n = int(rng.choice([1, 7, 8, 9, 63, 500, 4096, 19999, 20000, 20480, 20481, 30000])) if i % 3 else int(rng.integers(1, 20481))
        ids = np.empty(n, np.int32)
        pos = 0
        mean_run = int(rng.choice([2, 11, 60, 700, 5000]))
        while pos < n:
            ln = min(n - pos, 1 + int(rng.geometric(1.0 / mean_run)))
            if rng.random() < 0.5:
                ids[pos:pos + ln] = rng.integers(0, dn)
            else:
                ids[pos:pos + ln] = rng.integers(0, dn, ln)
            pos += ln
        if i % 41 == 0:
            ids[int(rng.integers(0, n))] = dn + int(rng.integers(0, 5))  # beyond the dictionary
        page = orc.dict_encode_ids(ids, max(dn, int(ids.max()) + 1))
        if i % 29 == 0 and len(page) > 3:
            page = page[:len(page) - int(rng.integers(1, min(len(page) - 1, 40)))]  # stream cut short
        pages.append(gpu.Page(page, n, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic))
        expect.append(orc.dict_decode(page, dic, n))
    res = gpu.decode_pages(pages)
    for i, ((st, written, consumed, out, guard), (ost, ovals)) in enumerate(zip(res, expect)):
        assert st == ost and written == ovals.size, (i, st, ost, written, ovals.size, pages[i].num_values)
        assert (out.view(ut)[:written] == ovals.view(ut)).all(), i
        assert (guard == 0xA5).all(), i


# ------------------------------------------------------------------------------- delta
@pytest.mark.parametrize("elem", [4, 8])
def test_delta_pages(gpu, orc, elem):
    import parquet_java_amd as pq
    from javarand import JavaRandom
    from test_oracle_kat import delta_cases_long
    dt = np.int64 if elem == 8 else np.int32
    pages, expect = [], []

    def add(page, cap):
        pages.append(gpu.Page(page, cap, pq.ENC_DELTA_BINARY_PACKED, elem))
        expect.append(orc.delta_decode(page, cap))

    for name, data in delta_cases_long():
        v = np.array(data, np.int64)
        if elem == 4:
            v = v.astype(np.int32)
        add(orc.delta_encode(v.astype(dt)), max(len(data), 100))
    r = JavaRandom(0)
    for _ in range(40):  # randomDataTest shape
        size = r.next_int(1000)
        v = np.array([r.next_long() for _ in range(size)], np.int64).astype(dt)
        add(orc.delta_encode(v), 1000)
    rng = np.random.default_rng(3)
    for k in (0, 4, 12, 20, 31, 40, 63):  # timestamp-like columns, widths ~k
        if elem == 4 and k > 31:
            continue
        n = 20_000 - 3
        step = rng.integers(0, 1 << k, n, dtype=np.uint64).astype(np.int64) + 1000 if k else np.full(n, 1000, np.int64)
        ts = (np.int64(1_600_000_000_000_000_000 if elem == 8 else 1000) + np.cumsum(step)).astype(dt)
        add(orc.delta_encode(ts), n)
    for seed in (1, 2, 3):  # block lengths that change every few blocks: widths by regime, min-delta varints by a byte
        r2 = np.random.default_rng(100 + seed)
        parts = [r2.integers(0, 1 << int(k), int(m), dtype=np.uint64).astype(np.int64) + int(off)
                 for k, m, off in zip(r2.integers(1, 30 if elem == 8 else 20, 40), r2.integers(100, 900, 40), r2.integers(0, 20000, 40))]
        ts = (np.int64(1 << 40 if elem == 8 else 1000) + np.cumsum(np.concatenate(parts))).astype(dt)
        add(orc.delta_encode(ts), ts.size)
    for block, mini in ((256, 8), (64, 1), (1024, 4), (2048, 16)):  # other configurations
        v = rng.integers(-(1 << 40), 1 << 40, 5000).astype(dt)
        add(orc.delta_encode(v, block, mini), 5000)
    big = np.cumsum(rng.integers(-1000, 1000, 300_000)).astype(dt)  # many windows
    add(orc.delta_encode(big), 300_000)
    # malformed
    good = orc.delta_encode(np.arange(1000, dtype=dt) * 3)
    for page, cap in ((good[:3], 10), (good[:200], 1000), (good, 999), (bytes([100, 3, 3, 0]), 10),
                      (bytes([0x80, 0x01, 4, 40, 0, 0, 65, 0, 0, 0]), 100), (bytes([0x80, 0x01, 4, 40, 0]), 100), (b"", 5),
                      (bytes([0x80, 0x01, 4, 100, 14, 2, 8, 0, 0, 0, 1, 2]), 100)):
        add(page, cap)
    res = gpu.decode_pages(pages)
    for i, ((st, written, consumed, out, guard), (ost, ovals, ototal, oused)) in enumerate(zip(res, expect)):
        assert st == ost, (i, st, ost)
        if ost == 0:
            assert written == ototal and consumed == oused, (i, written, ototal, consumed, oused)
            assert (out.view(dt)[:written] == ovals.astype(dt)).all(), i
        else:
            assert written == 0
        assert (guard == 0xA5).all()


# ------------------------------------------------------------------------------- byte stream split, BIT_PACKED
def test_bss_pages(gpu, orc):
    import parquet_java_amd as pq
    rng = np.random.default_rng(13)
    pages, expect = [], []
    for k in (4, 8, 1, 2, 3, 12, 16):
        for n in (0, 1, 3, 4, 5, 255, 256, 20_000, 100_001):
            raw = rng.integers(0, 256, n * k, dtype=np.uint8)
            enc = orc.bss_encode(raw, k)
            pages.append(gpu.Page(enc, n + (n % 3), pq.ENC_BYTE_STREAM_SPLIT, k))
            expect.append(orc.bss_decode(enc, k, n + (n % 3)))
    for enc, k, bound in ((np.zeros(7, np.uint8), 4, 10), (np.zeros(16, np.uint8), 4, 3)):
        pages.append(gpu.Page(enc, bound, pq.ENC_BYTE_STREAM_SPLIT, k))
        expect.append(orc.bss_decode(enc, k, bound))
    # the reference's known-answer float vector
    kat = bytes([0x40, 0x00, 0x80, 0x40, 0x05, 0x84, 0xC5, 0xBD, 0x32, 0xC2, 0x41, 0x42])
    pages.append(gpu.Page(kat, 3, pq.ENC_BYTE_STREAM_SPLIT, 4))
    expect.append(orc.bss_decode(kat, 4, 3))
    # decode one launch per K (the launcher takes mixed K, the harness groups by elem size)
    by_k = {}
    for i, p in enumerate(pages):
        by_k.setdefault(p.elem_size, []).append(i)
    for k, idx in by_k.items():
        res = gpu.decode_pages([pages[i] for i in idx])
        for i, (st, written, consumed, out, guard) in zip(idx, res):
            ost, oout = expect[i]
            assert st == ost, (k, i)
            if ost == 0:
                assert written == oout.size // k and consumed == pages[i].data.size
                assert (out[:oout.size] == oout).all(), (k, i)
            assert (guard == 0xA5).all()
    assert pages[-1] and np.frombuffer(kat, np.uint8).size == 12


def test_bitpacked_levels(gpu, orc):
    import parquet_java_amd as pq
    from test_oracle_kat import BE_KATS, bits
    rng = np.random.default_rng(17)
    pages, expect = [], []
    for w, vals, s in BE_KATS:
        pages.append(gpu.Page(bits(s), len(vals), pq.ENC_BIT_PACKED, 4, bit_width=w))
        expect.append(orc.bitpacked_levels_decode(w, bits(s), len(vals)))
    for w in range(0, 33):
        for n in (1, 8, 13, 1000):
            vals = rand_vals(rng, w, (n + 7) // 8 * 8)
            enc = orc.pack(w, vals, msb_first=True) if w else np.zeros(0, np.uint8)
            for data in (enc, enc[:max(0, enc.size - 2)]):  # complete, and clipped by the stream end
                pages.append(gpu.Page(data, n, pq.ENC_BIT_PACKED, 4, bit_width=w))
                expect.append(orc.bitpacked_levels_decode(w, data, n))
    res = gpu.decode_pages(pages)
    for i, ((st, written, consumed, out, guard), (ost, ovals, oused)) in enumerate(zip(res, expect)):
        assert st == ost == 0 and written == ovals.size and consumed == oused, i
        assert (out.view(np.int32) == ovals).all(), i


# ------------------------------------------------------------------------------- host-memory entry points
def test_host_entry_points_and_reader_mirror(gpu, orc):
    import parquet_java_amd as pq
    rng = np.random.default_rng(23)
    # ParquetReadRouter on the reference's own test input shape (TestParquetReadRouter.java:41-57: byte i = (byte) i)
    data = np.arange(2048, dtype=np.uint64).astype(np.uint8)
    for w in range(1, 33):
        n = (data.size * 8 // w) // 8 * 8
        for method in (pq.ParquetReadRouter.read, pq.ParquetReadRouter.readBatch, pq.ParquetReadRouter.readBatchUsing512Vector):
            s = pq.ByteBufferInputStream.wrap(data)
            out = np.zeros(n, np.int32)
            method(w, s, n, out)
            assert s.position() == n * w // 8
            assert (out == orc.read_batch(w, data, n)[1]).all()
        with pytest.raises(pq.EOFException):
            pq.ParquetReadRouter.read(w, pq.ByteBufferInputStream.wrap(data[:10]), 2048, np.zeros(2048, np.int32))
    # dictionary reader
    dic = rng.integers(-1 << 62, 1 << 62, 50).astype(np.int64)
    ids = np.array([i % 50 for i in range(2000)], np.int32)
    rd = pq.DictionaryValuesReader(dic)
    rd.initFromPage(2000, pq.ByteBufferInputStream.wrap(orc.dict_encode_ids(ids, 50)))
    assert [rd.readLong() for _ in range(2000)] == dic[ids].tolist()
    with pytest.raises(pq.IllegalArgumentException):
        rd.readLong()
    # delta reader reports the offset, like shouldReturnCorrectOffsetAfterInitialization
    v = np.arange(2 * 128 + 3, dtype=np.int64) * 32
    content = orc.delta_encode(v)
    page = np.zeros(content.size * 10, np.uint8)
    page[33:33 + content.size] = content
    s = pq.ByteBufferInputStream.wrap(page)
    s.skipFully(33)
    rd = pq.DeltaBinaryPackingValuesReader()
    rd.initFromPage(100 + v.size, s)
    assert s.position() == content.size + 33
    assert [rd.readLong() for _ in range(v.size)] == v.tolist()
    with pytest.raises(pq.ParquetDecodingException):
        rd.readLong()
    # byte stream split floats
    rd = pq.ByteStreamSplitValuesReader(np.float32)
    rd.initFromPage(3, pq.ByteBufferInputStream.wrap(
        bytes([0x40, 0x00, 0x80, 0x40, 0x05, 0x84, 0xC5, 0xBD, 0x32, 0xC2, 0x41, 0x42])))
    assert [rd.readFloat() for _ in range(3)] == [-98.62548828125, 23.62744140625, 44.62939453125]
    with pytest.raises(pq.ParquetDecodingException):
        rd.readFloat()
    # V1 levels
    body = orc.rle_encode(2, [1, 2, 3, 0] * 10)
    s = pq.ByteBufferInputStream.wrap(struct.pack("<i", body.size) + bytes(body) + b"\x07")
    rd = pq.RunLengthBitPackingHybridValuesReader(2)
    rd.initFromPage(40, s)
    assert [rd.readInteger() for _ in range(40)] == [1, 2, 3, 0] * 10 and s.available() == 1


def test_byte_packer_mirror(gpu, orc):
    """Per-step interface: Packer.LITTLE_ENDIAN / BIG_ENDIAN .newBytePacker(w).unpack8Values /
    unpack32Values, the plugin's unpackValuesUsingVector == the scalar packer for every width
    (TestByteBitPacking512VectorLE.java:60-87), BytePackerForLong (TestByteBitPacking.java:149-198)."""
    import parquet_java_amd as pq
    rng = np.random.default_rng(31)
    for w in range(0, 33):
        raw = rng.integers(0, 256, 3 + 64 * 4 + 8, dtype=np.uint8)
        for packer, msb in ((pq.Packer.LITTLE_ENDIAN.newBytePacker(w), False), (pq.Packer.BIG_ENDIAN.newBytePacker(w), True)):
            assert packer.getBitWidth() == w
            out = np.full(48, -7, np.int32)
            packer.unpack8Values(raw, 3, out, 5)
            exp = orc.unpack8(w, raw[3:3 + w], msb_first=msb) if w else np.zeros(8, np.int32)
            assert (out[5:13] == exp).all() and (out[:5] == -7).all() and (out[13:] == -7).all(), (w, msb)
            packer.unpack32Values(raw.tobytes(), 1, out, 16)
            exp32 = np.concatenate([orc.unpack8(w, raw[1 + g * w:1 + (g + 1) * w], msb_first=msb) for g in range(4)]) \
                if w else np.zeros(32, np.int32)
            assert (out[16:48] == exp32).all(), (w, msb)
        vec = pq.Packer.LITTLE_ENDIAN.newBytePackerVector(w)
        cnt = vec.getUnpackCount()
        assert cnt == (0 if w == 0 else 64 if w == 8 else 32 if w < 16 else 16)
        out = np.full(cnt + 2, -7, np.int32)
        vec.unpackValuesUsingVector(raw, 2, out, 1)
        scalar = np.zeros(max(cnt, 1), np.int32)
        sp = pq.Packer.LITTLE_ENDIAN.newBytePacker(w)
        for g in range(cnt // 8):
            sp.unpack8Values(raw, 2 + g * w, scalar, g * 8)
        assert (out[1:1 + cnt] == scalar[:cnt]).all() and out[0] == -7 and out[-1] == -7, w
    with pytest.raises(RuntimeError):
        pq.Packer.BIG_ENDIAN.newBytePackerVector(3)
    with pytest.raises(IndexError):
        pq.Packer.LITTLE_ENDIAN.newBytePacker(9).unpack8Values(np.zeros(8, np.uint8), 0, np.zeros(8, np.int32), 0)
    for w in (0, 1, 7, 31, 33, 47, 64):
        raw = rng.integers(0, 256, 2 + 4 * w, dtype=np.uint8)
        p = pq.Packer.LITTLE_ENDIAN.newBytePackerForLong(w)
        out = np.zeros(40, np.int64)
        p.unpack8Values(raw, 2, out, 0)
        p.unpack32Values(raw, 2, out, 8)
        exp = np.concatenate([orc.unpack8(w, raw[2 + g * w:2 + (g + 1) * w], long=True) for g in range(4)]) if w else np.zeros(32, np.int64)
        assert (out[:8] == exp[:8]).all() and (out[8:] == exp).all(), w


def hybrid_stream(orc, w, runs):
    """Hand-built hybrid stream.  runs: ('p', groups, values[groups*8]) or ('r', count, value)."""
    out = bytearray()

    def varint(h):
        while h >= 0x80:
            out.append((h & 0x7F) | 0x80)
            h >>= 7
        out.append(h)
    for r in runs:
        if r[0] == 'p':
            varint((r[1] << 1) | 1)
            out.extend(orc.pack(w, np.asarray(r[2], np.int32)).tobytes())
        else:
            varint(r[1] << 1)
            out.extend(int(r[2]).to_bytes((w + 7) // 8, "little"))
    return np.frombuffer(bytes(out), np.uint8).copy()


def test_rle_long_regular_pages(gpu, orc, rle_path):
    """Pages that start with many equal bit-packed runs (planned in bulk) in every shape around the
    bulk path's edges: run lengths other than 63 groups, multi-byte headers, an irregular run part
    way, a cut-off stream, fewer values wanted than the stream holds."""
    rng = np.random.default_rng(77)
    cases = []

    def regular(w, groups, n_runs):
        return [('p', groups, rand_vals(rng, w, groups * 8)) for _ in range(n_runs)]
    for w, groups, n_runs in ((8, 63, 400), (3, 63, 600), (16, 63, 140), (32, 1, 9000), (24, 1, 20000), (5, 200, 100),
                              (12, 2, 9000), (1, 700, 40), (7, 64, 300)):
        runs = regular(w, groups, n_runs)
        total = groups * 8 * n_runs
        data = hybrid_stream(orc, w, runs)
        cases.append((w, data, total, 0))
        cases.append((w, data, total - 5, 0))               # the last run is not wholly wanted
        cases.append((w, data, total + 9, 0))               # reads past the end of the stream
        cases.append((w, data[:data.size - 7], total, 0))   # the last run is cut short
        cases.append((w, data[:data.size * 2 // 3], total, 0))
        # an RLE run, a shorter packed run and a packed run with another header part way
        mid = n_runs * 2 // 3
        mixed = runs[:mid] + [('r', 100, 1)] + runs[mid:mid + 3] + [('p', max(groups - 1, 1), rand_vals(rng, w, max(groups - 1, 1) * 8))] + runs[mid + 3:]
        tot2 = total + 100 + max(groups - 1, 1) * 8
        cases.append((w, hybrid_stream(orc, w, mixed), tot2, 0))
        cases.append((w, hybrid_stream(orc, w, mixed), tot2 // 2, 0))
    # V1 levels and ids with their prefixes
    import parquet_java_amd as pq
    runs = regular(2, 63, 300)
    body = hybrid_stream(orc, 2, runs)
    cases.append((2, np.concatenate([np.frombuffer(struct.pack("<i", body.size), np.uint8), body, np.full(9, 0xEE, np.uint8)]),
                  63 * 8 * 300, pq.F_LENGTH_PREFIX))
    body = hybrid_stream(orc, 11, regular(11, 63, 250))
    cases.append((0, np.concatenate([np.array([11], np.uint8), body]), 63 * 8 * 250, pq.F_WIDTH_PREFIX))
    check_rle(gpu, orc, cases)


@pytest.mark.parametrize("elem", [4, 8])
def test_dictionary_long_regular_pages(gpu, orc, elem, dict_path):
    import parquet_java_amd as pq
    rng = np.random.default_rng(78)
    pages, expect = [], []
    for w, groups, n_runs, cut in ((10, 63, 300, 0), (16, 63, 130, 11), (4, 63, 500, 0), (12, 7, 2000, 3)):
        dic = rng.integers(-(1 << 31), 1 << 31, 1 << w).astype(np.int32 if elem == 4 else np.int64)
        if elem == 8:
            dic = dic * np.int64(0x100000001)
        runs = [('p', groups, rand_vals(rng, w, groups * 8)) for _ in range(n_runs)]
        runs.insert(n_runs // 2, ('r', 4000, 5))
        data = np.concatenate([np.array([w], np.uint8), hybrid_stream(orc, w, runs)])
        n = groups * 8 * n_runs + 4000 - cut
        pages.append(gpu.Page(data, n, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic))
        expect.append(orc.dict_decode(data, dic, n))
    for (st, written, _, out, guard), (ost, ovals) in zip(gpu.decode_pages(pages), expect):
        assert st == ost == 0 and written == ovals.size
        assert (out.view(ovals.dtype) == ovals).all() and (guard == 0xA5).all()


@pytest.mark.parametrize("limit", [0, 3, 7, 12, 40])
def test_pages_that_find_the_chunk_pool_full(gpu, orc, monkeypatch, limit):
    """With fewer pool chunks than the pages need (PQD_POOL_CHUNKS), the pages that find no room
    are decoded by the generic kernel; results do not change.  Long regular pages ask for many
    chunks at once (bulk path), irregular ones chunk by chunk."""
    import parquet_java_amd as pq
    monkeypatch.setenv("PQD_POOL_CHUNKS", str(limit))
    monkeypatch.setenv("PQD_RLE_PATH", "planned")
    monkeypatch.setenv("PQD_DICT_PATH", "sliced")
    rng = np.random.default_rng(91 + limit)
    cases = []
    for w, n in ((8, 300_000), (3, 70_000), (12, 5000), (16, 131_072)):
        cases.append((w, orc.rle_encode(w, rand_vals(rng, w, n)), n, 0))
    runs = np.repeat(rng.integers(0, 16, 6000), rng.geometric(1 / 30.0, 6000))[:150_000].astype(np.int32)
    cases.append((4, orc.rle_encode(4, runs), runs.size, 0))
    cases.append((5, orc.rle_encode(5, rand_vals(rng, 5, 100)), 100, 0))
    check_rle(gpu, orc, cases)
    for elem, dt in ((4, np.int32), (8, np.int64)):
        dic = rng.integers(-(1 << 31), 1 << 31, 1 << 10).astype(dt)
        pages, expect = [], []
        for n in (100_000, 20_000, 50, 45_000):
            ids = rng.integers(0, dic.size, n).astype(np.int32)
            data = orc.dict_encode_ids(ids, dic.size)
            pages.append(gpu.Page(data, n, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic))
            expect.append(dic[ids])
        for (st, written, _, out, guard), exp in zip(gpu.decode_pages(pages), expect):
            assert st == 0 and written == exp.size and (out.view(dt) == exp).all() and (guard == 0xA5).all()


@pytest.mark.parametrize("elem", [4, 8])
def test_dictionary_rle_value_wider_than_the_id_width(gpu, orc, elem, dict_path):
    """The value of an RLE run is ceil(w/8) bytes and is NOT masked to w bits
    (BytesUtils.readIntLittleEndianPaddedOnBitWidth, BytesUtils.java:124-142): a page of 3-bit ids
    can refer to entry 200.  Valid with 300 entries, out of range with 100."""
    import parquet_java_amd as pq
    dt = np.int32 if elem == 4 else np.int64
    packed = orc.pack(3, np.arange(8, dtype=np.int32) % 8)
    page = np.concatenate([np.array([3, (1 << 1) | 1], np.uint8), packed, np.array([10 << 1, 200, 2 << 1, 5], np.uint8)])
    res = []
    for entries in (300, 100):
        dic = (np.arange(entries) * 7 - 3).astype(dt)
        p = gpu.Page(page, 20, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic)
        (st, written, _, out, guard), = gpu.decode_pages([p])
        ost, ovals = orc.dict_decode(page, dic, 20)
        assert st == ost and written == ovals.size and (out.view(dt)[:written] == ovals).all() and (guard == 0xA5).all()
        res.append((st, written))
    assert res == [(0, 20), (pq.DICT_OOB, 8)]


@pytest.mark.parametrize("slots", [None, "1", "0"])
def test_eight_byte_dictionaries_of_every_size_class(gpu, orc, monkeypatch, slots):
    """8-byte dictionaries are looked up in dword planes, a quarter (16384 entries) of a plane at a time:
    one quarter per plane (both resident), two, three and four quarters, the quarter boundaries, a
    dictionary longer than 16-bit ids can reach; several dictionaries in one call, with room for all of
    their plane copies, for one of them (the others go to the generic kernel) and for none."""
    import parquet_java_amd as pq
    if slots is not None:
        monkeypatch.setenv("PQD_PLANE_SLOTS", slots)
    rng = np.random.default_rng(99)
    pages, expect = [], []
    for n_dict in (1, 2, 5, 8191, 8192, 8193, 16383, 16384, 16385, 20000, 32767, 32768, 32769, 40000, 49152, 49153, 65535, 65536, 70000):
        dic = rng.integers(-(1 << 63), (1 << 63) - 1, n_dict, dtype=np.int64)
        reach = min(n_dict, 65536)
        for n in (20000, 777):
            ids = rng.integers(0, reach, n).astype(np.int32)
            ids[:8] = [0, reach - 1, reach // 2, max(reach // 4 - 1, 0), reach // 4, max(3 * reach // 4 - 1, 0), 3 * reach // 4, reach - 1]
            ids[8:16] = np.minimum([0, 16383, 16384, 32767, 32768, 49151, 49152, 65535], reach - 1)
            if n == 777:
                ids[400:] = ids[3]  # an RLE run
            pages.append(gpu.Page(orc.dict_encode_ids(ids, n_dict), n, pq.ENC_RLE_DICTIONARY, 8, dictionary=dic))
            expect.append(dic[ids])
    for (st, written, _, out, guard), want in zip(gpu.decode_pages(pages), expect):
        assert st == 0 and written == want.size
        assert (out.view(np.int64) == want).all()
        assert (guard == 0xA5).all()


@pytest.mark.parametrize("elem", [4, 8])
@pytest.mark.parametrize("mode", ["only", "0"])
def test_dictionary_batches_of_regular_chunks(gpu, orc, monkeypatch, elem, mode):
    """Batches in which every chunk is nothing but whole bit-packed runs at one distance (non-repeating ids as
    parquet-java writes them) over dictionaries of more than 49152 entries take a path of their own inside
    the sliced gather.  PQD_DICT_REGULAR=only lets nothing else write (a batch that did not qualify would
    come back unwritten), =0 keeps the general path: both must give the oracle's values.  Page lengths that
    end inside a unit, span chunks (the second chunk begins inside a run), take the bulk plan path or are a
    single run; narrow ids over a large dictionary; several dictionaries and an empty page in one call;
    dictionaries that end inside the last quarter, with an id beyond them."""
    import parquet_java_amd as pq
    monkeypatch.setenv("PQD_DICT_PATH", "sliced")
    monkeypatch.setenv("PQD_DICT_REGULAR", mode)
    dt = np.int32 if elem == 4 else np.int64
    rng = np.random.default_rng(4100 + elem)

    def non_repeating(n, reach):
        ids = rng.integers(0, reach, n).astype(np.int32)
        same = np.flatnonzero(ids[1:] == ids[:-1]) + 1  # no two equal neighbours: the encoder never sees 8 repeats
        ids[same] = (ids[same] + 1) % reach
        return ids

    pages, expect = [], []
    for n_dict in (65536, 49153, 57000, 70000):
        dic = rng.integers(-(1 << 31), 1 << 31, n_dict).astype(dt)
        if elem == 8:
            dic = dic * np.int64(0x100000001) + 7
        reach = min(n_dict, 65536)
        for n in (20000, 19997, 504 * 7, 1000, 8, 3, 40000, 100_000, 0):
            ids = non_repeating(n, reach)
            if n >= 16:
                ids[:8] = [0, reach - 1, 16383, 16384, 32767, 32768, 49151, 49152]
            data = orc.dict_encode_ids(ids, n_dict) if n else np.array([16], np.uint8)
            pages.append(gpu.Page(data, n, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic))
            expect.append((0, dic[ids]))
        for w in (3, 10, 13):  # pages written while the dictionary was still small
            ids = non_repeating(20000, 1 << w)
            data = np.concatenate([np.array([w], np.uint8), orc.rle_encode(w, ids)])
            pages.append(gpu.Page(data, ids.size, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic))
            expect.append((0, dic[ids]))
        if n_dict < 65536:  # ids of 16 bits can point beyond such a dictionary
            ids = non_repeating(20000, n_dict)
            ids[12345] = n_dict + 5
            ids[15000] = 65535
            data = orc.dict_encode_ids(ids, n_dict)
            pages.append(gpu.Page(data, ids.size, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic))
            ost, ovals = orc.dict_decode(data, dic, ids.size)
            assert ost == pq.DICT_OOB and ovals.size == 12345
            expect.append((ost, ovals))
    res = gpu.decode_pages(pages)
    for i, ((st, written, _, out, guard), (ost, ovals)) in enumerate(zip(res, expect)):
        assert st == ost and written == ovals.size, (i, st, written, ost, ovals.size)
        assert (out.view(dt)[:written] == ovals).all(), i
        assert (guard == 0xA5).all(), i
    # a few irregular chunks (here: pages with an RLE run) in a large batch go to the left-over kernel's gather from
    # memory and the batch keeps the regular path: one in sixteen pages at most, 5 of these 96
    dic = rng.integers(-(1 << 31), 1 << 31, 65536).astype(dt)
    many, want = [], []
    for i in range(96):
        ids = non_repeating(6000 + 13 * i, 65536)
        if i in (7, 40, 41, 90):
            ids[1000:1100] = 4242
        many.append(gpu.Page(orc.dict_encode_ids(ids, 65536), ids.size, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic))
        want.append(dic[ids])
    monkeypatch.setenv("PQD_DIRECT_MAX", "4096")  # (the pages themselves are longer than that: only the budget sends chunks away)
    for i, ((st, written, _, out, guard), w_) in enumerate(zip(gpu.decode_pages(many), want)):
        assert st == 0 and written == w_.size and (out.view(dt) == w_).all() and (guard == 0xA5).all(), i
    monkeypatch.delenv("PQD_DIRECT_MAX")
    if mode == "only":  # the switch does what the test relies on: a batch with an irregular chunk that is not listed is not written at all
        monkeypatch.setenv("PQD_DIRECT_MAX", "0")  # (no chunk is handed to the left-over launch)
        dic = rng.integers(-(1 << 31), 1 << 31, 65536).astype(dt)
        ids = non_repeating(20000, 65536)
        ids[5000:5100] = 77  # an RLE run
        bad = [gpu.Page(orc.dict_encode_ids(non_repeating(20000, 65536), 65536), 20000, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic),
               gpu.Page(orc.dict_encode_ids(ids, 65536), 20000, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic)]
        for st, written, _, out, guard in gpu.decode_pages(bad):
            assert st == 0 and written == 20000 and (out == 0xA5).all()


@pytest.mark.parametrize("elem", [4, 8])
@pytest.mark.parametrize("direct_max", ["8192", "0", "3000"])
@pytest.mark.parametrize("pool", [None, "9"])
def test_dictionary_short_chunks_gathered_from_memory(gpu, orc, monkeypatch, elem, direct_max, pool):
    """Chunks of at most PQD_DIRECT_MAX values (default 8192) over a dictionary that would have to be swept through
    LDS are taken from the end of the chunk pool and expanded by the left-over kernel, which reads the dictionary
    from memory: run-heavy pages (cut into several chunks of a few thousand values), small pages, a page's last
    chunk.  With the switch at 0 everything goes through the sliced gather; with a pool of 9 chunks most pages
    find no room and are decoded by the generic kernel.  All must give the oracle's values, ids beyond the
    dictionary included."""
    import parquet_java_amd as pq
    monkeypatch.setenv("PQD_DICT_PATH", "sliced")
    monkeypatch.setenv("PQD_DIRECT_MAX", direct_max)
    monkeypatch.setenv("PQD_WHOLE_MAX", "0")  # (run-heavy pages would otherwise be decoded whole and never be cut into chunks)
    if pool is not None:
        monkeypatch.setenv("PQD_POOL_CHUNKS", pool)
    dt = np.int32 if elem == 4 else np.int64
    rng = np.random.default_rng(5200 + elem)
    n_dict = 65536
    dic = rng.integers(-(1 << 31), 1 << 31, n_dict).astype(dt)
    if elem == 8:
        dic = dic * np.int64(0x100000001) - 11
    short = rng.integers(-(1 << 31), 1 << 31, 40000).astype(dt)  # 16-bit ids can point beyond this one
    pages, expect = [], []

    def add(ids, d):
        data = orc.dict_encode_ids(ids, d.size)
        pages.append(gpu.Page(data, ids.size, pq.ENC_RLE_DICTIONARY, elem, dictionary=d))
        expect.append(orc.dict_decode(data, d, ids.size))

    for n in (5000, 20000, 8192, 8193, 100, 7):  # small pages and the last chunk of longer ones
        add(rng.integers(0, n_dict, n).astype(np.int32), dic)
    for _ in range(4):  # run-heavy pages: several chunks each
        x = np.sort(rng.integers(0, n_dict, 600)).repeat(rng.integers(1, 70, 600))[:20000].astype(np.int32)
        add(x, dic)
    x = np.repeat(rng.integers(0, n_dict, 3000), rng.integers(1, 20, 3000))[:30000].astype(np.int32)  # short runs, unsorted
    add(x, dic)
    ids = rng.integers(0, short.size, 6000).astype(np.int32)
    ids[4321] = 50000  # beyond the dictionary, inside a short chunk
    add(ids, short)
    x = np.sort(rng.integers(0, short.size, 500)).repeat(rng.integers(1, 80, 500))[:20000].astype(np.int32)
    x[15000:15010] = 65535
    add(x, short)
    for i, ((st, written, _, out, guard), (ost, ovals)) in enumerate(zip(gpu.decode_pages(pages), expect)):
        assert st == ost and written == ovals.size, (i, st, written, ost, ovals.size)
        assert (out.view(dt)[:written] == ovals).all(), i
        assert (guard == 0xA5).all(), i


@pytest.mark.parametrize("elem", [4, 8])
def test_dictionary_run_heavy_pages_decoded_whole(gpu, orc, monkeypatch, elem):
    """Dictionary pages whose stream is short for their number of values (at most 8 KiB and at most half of the
    bit-packed size: sorted or clustered ids) are not planned: the left-over launch parses the whole stream in LDS
    and looks the ids up in one pass.  Widths from 1 to 21 bits, one run of 20000 values behind a three-byte
    header, the largest page the path takes and the first it does not, ids beyond the dictionary in an RLE run and
    in a bit-packed group, an RLE value wider than the page's bit width (the reference does not mask it), and
    streams the path must hand to the page decoder (a zero-length run, a cut-off run, a stream that ends short of
    the page).  PQD_WHOLE_MAX=0 switches the path off: statuses, counts, consumed bytes and values must agree with
    that run and with the oracle."""
    import parquet_java_amd as pq
    dt = np.int32 if elem == 4 else np.int64
    rng = np.random.default_rng(6300 + elem)
    pages, expect = [], []

    def dictionary(n):
        d = rng.integers(-(1 << 31), 1 << 31, n).astype(dt)
        return d * np.int64(0x100000001) + 3 if elem == 8 else d

    def add(data, n, d):
        data = np.asarray(data, np.uint8)
        pages.append(gpu.Page(data, n, pq.ENC_RLE_DICTIONARY, elem, dictionary=d))
        expect.append(orc.dict_decode(data, d, n))

    def runs(n, reach, n_runs, longest):
        x = np.sort(rng.integers(0, reach, n_runs)).repeat(rng.integers(1, longest, n_runs))[:n].astype(np.int32)
        return np.concatenate([x, np.full(n - x.size, x[-1], np.int32)])

    for n_dict in (2, 30, 4096, 65536, (1 << 20) + 1):
        d = dictionary(n_dict)
        for n in (20000, 32768, 32769, 1000, 7):
            ids = runs(n, n_dict, max(n // 33, 1), 70)
            add(orc.dict_encode_ids(ids, n_dict), n, d)
        add(orc.dict_encode_ids(np.full(20000, n_dict - 1, np.int32), n_dict), 20000, d)  # one run, three header bytes
        add(orc.dict_encode_ids(rng.permutation(runs(20000, n_dict, 300, 140)), n_dict), 20000, d)  # not run-heavy at all
    d = dictionary(40000)
    ids = runs(20000, 40000, 600, 70)
    ids[15000:15020] = 50000  # beyond the dictionary: an RLE run
    add(orc.dict_encode_ids(ids, 40000), 20000, d)
    ids = runs(20000, 40000, 600, 70)
    ids[12000:12003] = (1, 65535, 2)  # inside a bit-packed group
    add(orc.dict_encode_ids(ids, 40000), 20000, d)
    # written with 12-bit ids; an RLE run whose two value bytes say 0x9C40 = 40000 (beyond) and 0x1234 (inside, above 2^12)
    body = orc.rle_encode(12, runs(8000, 4096, 200, 80))
    for value, n in ((0x9C40, 8600), (0x1234, 8600)):
        add(np.concatenate([np.array([12], np.uint8), body, np.array([600 << 1 & 0x7F | 0x80, 600 >> 6, value & 0xFF, value >> 8], np.uint8)]), n, d)
    # streams of more than one 8 KiB window: short runs (about 14 KB); long bit-packed runs that cross the windows' ends
    # between stretches of RLE runs; the same with an id beyond the dictionary in the third window and cut short there
    big = dictionary(65536)
    x = np.repeat(rng.integers(0, 65536, 2400), rng.integers(1, 21, 2400))[:20000].astype(np.int32)
    add(orc.dict_encode_ids(np.concatenate([x, np.full(20000 - x.size, x[-1], np.int32)]), 65536), 20000, big)
    mixed = np.concatenate([runs(6000, 65536, 150, 80), rng.integers(0, 65536, 5000), runs(8000, 65536, 200, 80),
                            rng.integers(0, 65536, 5000), runs(8768, 65536, 200, 90)]).astype(np.int32)
    page = orc.dict_encode_ids(mixed, 65536)
    assert 16384 < page.size <= 32768
    add(page, mixed.size, big)
    add(page, 20001, big)  # the page ends inside the stream
    mixed2 = mixed.copy()
    mixed2[25000:25012] = 50000
    page2 = orc.dict_encode_ids(mixed2, 65536)
    add(page2, mixed2.size, d)  # 40000 entries: beyond the dictionary
    add(page2[:20000], mixed2.size, d)
    good = orc.dict_encode_ids(runs(20000, 40000, 600, 70), 40000)
    add(good[:-1], 20000, d)  # the last run is cut off
    add(good[: good.size // 2], 20000, d)  # the stream ends short of the page
    add(np.concatenate([good[:1], np.array([0, 7, 0], np.uint8), good[1:]]), 20000, d)  # a run of zero values in front
    add(good, 20000, d)
    results = {}
    for mode in ("0", None):
        if mode is None:
            monkeypatch.delenv("PQD_WHOLE_MAX")
        else:
            monkeypatch.setenv("PQD_WHOLE_MAX", mode)
        results[mode] = gpu.decode_pages(pages)
        for i, ((st, written, _, out, guard), (ost, ovals)) in enumerate(zip(results[mode], expect)):
            assert st == ost and written == ovals.size, (mode, i, st, written, ost, ovals.size)
            assert (out.view(dt)[:written] == ovals).all(), (mode, i)
            assert (guard == 0xA5).all(), (mode, i)
    for i, (a, b) in enumerate(zip(results["0"], results[None])):
        assert a[2] == b[2], (i, a[2], b[2])  # bytes consumed


@pytest.mark.parametrize("kind", ["plain", "dict4", "dict8"])
def test_long_run_heavy_pages_decoded_whole(gpu, orc, monkeypatch, kind):
    """Pages of more than 32768 values (or 32 KiB of stream) made of many short runs are not planned either: the
    left-over launch takes them window by window, a window ending at its 32768th value (inside an RLE run, whose
    remaining values are then stored directly, or in front of a bit-packed run).  Plain hybrid pages with each kind of
    prefix and dictionary pages; RLE runs of several windows' worth of values, a page that ends inside such a run,
    bit-packed runs of more values than a window takes, ids beyond the dictionary in a later window and in the
    carried part of a run, streams cut short.  PQD_WHOLE_MAX=0 switches the path off: both runs must agree with the
    oracle and with each other in the bytes consumed."""
    import parquet_java_amd as pq
    rng = np.random.default_rng({"plain": 71, "dict4": 72, "dict8": 73}[kind])
    pages, expect = [], []
    elem = 8 if kind == "dict8" else 4
    dt = np.int64 if elem == 8 else np.int32

    def short_runs(n, w, mean):
        lens = rng.geometric(1.0 / mean, n // max(mean // 2, 1) + 16)
        vals = np.repeat(rng.integers(0, 1 << w, lens.size), lens)[:n]
        assert vals.size == n
        return vals.astype(np.uint32).view(np.int32)

    def add_plain(w, vals, n, prefix, cut=0):
        body = bytes(orc.rle_encode(w, vals))
        if cut:
            body = body[:len(body) - cut]
        if prefix == "length":
            data, flags, bw = len(body).to_bytes(4, "little") + body + b"\x99\x98", pq.F_LENGTH_PREFIX, w
            ost, ovals, _ = orc.rle_decode(w, data, n, prefixed=True)
        elif prefix == "width":
            data, flags, bw = bytes([w]) + body, pq.F_WIDTH_PREFIX, 0
            ost, ovals = orc.dict_decode_ids(data, n)
        else:
            data, flags, bw = body, 0, w
            ost, ovals, _ = orc.rle_decode(w, body, n)
        pages.append(gpu.Page(data, n, pq.ENC_RLE, 4, bit_width=bw, flags=flags))
        expect.append((ost, ovals))

    def add_dict(ids, n, d, cut=0):
        data = np.asarray(orc.dict_encode_ids(ids, d.size if d.size > 1 << 12 else 1 << 12), np.uint8)
        if cut:
            data = data[:data.size - cut]
        pages.append(gpu.Page(data, n, pq.ENC_RLE_DICTIONARY, elem, dictionary=d))
        expect.append(orc.dict_decode(data, d, n))

    if kind == "plain":
        for i, (w, n, mean) in enumerate([(1, 40000, 30), (3, 32769, 12), (8, 100_000, 40), (12, 70_001, 9), (16, 1_000_000, 40), (24, 50_000, 20)]):
            v = short_runs(n, w, mean)
            add_plain(w, v, n, ("none", "width", "length")[i % 3])
            add_plain(w, v, n + 20, ("width", "length", "none")[i % 3])     # reads past the encoded values
            add_plain(w, v, n - 1234, ("length", "none", "width")[i % 3])  # the page ends inside the stream
        w = 8
        v = short_runs(300_000, w, 25)
        v[50_000:250_000] = 77          # an RLE run of six windows' worth of values
        for n in (300_000, 180_000, 50_000 + 32768, 50_009):
            add_plain(w, v, n, "none")
        v = short_runs(200_000, w, 25)
        v[60_000:110_000] = rng.integers(0, 256, 50_000).astype(np.int32)  # bit-packed runs between the short ones
        add_plain(w, v, 200_000, "width")
        add_plain(w, v, 200_000, "none", cut=15_000)  # the stream ends short of the page
        add_plain(w, v, 200_000, "length", cut=1)
        # one bit-packed run of 40000 values behind a three-byte header, short runs around it
        a, b = short_runs(50_000, w, 25), short_runs(50_000, w, 25)
        mid = rng.integers(0, 256, 40_000).astype(np.int32)
        hdr, h = [], ((40_000 // 8) << 1) | 1
        while h >= 0x80:
            hdr.append((h & 0x7F) | 0x80)
            h >>= 7
        hdr.append(h)
        body = np.concatenate([orc.rle_encode(w, a), np.array(hdr, np.uint8), orc.pack(w, mid), orc.rle_encode(w, b)])
        ost, ovals, _ = orc.rle_decode(w, bytes(body), 140_000)
        pages.append(gpu.Page(bytes(body), 140_000, pq.ENC_RLE, 4, bit_width=w, flags=0))
        expect.append((ost, ovals))
    else:
        d = rng.integers(-(1 << 31), 1 << 31, 50_000).astype(dt)
        if elem == 8:
            d = d * np.int64(0x100000001) + 5
        for n, mean in ((40_000, 12), (250_000, 30), (1_000_000, 40)):
            ids = short_runs(n, 15, mean)
            add_dict(ids, n, d)
            add_dict(ids, n - 4321, d)
        ids = short_runs(200_000, 15, 20)
        ids[70_000:160_000] = 41_000           # a long run
        add_dict(ids, 200_000, d)
        add_dict(ids, 120_000, d)
        bad = ids.copy()
        bad[70_000:160_000] = 60_000           # ... beyond the dictionary
        add_dict(bad, 200_000, d)
        bad = ids.copy()
        bad[180_000:180_003] = (1, 65_000, 2)  # beyond the dictionary in a bit-packed group of a late window
        add_dict(bad, 200_000, d)
        add_dict(ids, 200_000, d, cut=9_000)   # the stream ends short of the page
    results = {}
    for mode in ("0", None):
        if mode is None:
            monkeypatch.delenv("PQD_WHOLE_MAX")
        else:
            monkeypatch.setenv("PQD_WHOLE_MAX", mode)
        results[mode] = gpu.decode_pages(pages)
        for i, ((st, written, _, out, guard), (ost, ovals)) in enumerate(zip(results[mode], expect)):
            assert st == ost and written == ovals.size, (mode, i, st, written, ost, ovals.size)
            assert (out.view(dt)[:written] == ovals).all(), (mode, i)
            assert (guard == 0xA5).all(), (mode, i)
    for i, (a, b) in enumerate(zip(results["0"], results[None])):
        assert a[2] == b[2], (i, a[2], b[2])  # bytes consumed
