"""Garbage in: random and half-random byte strings through the hybrid, dictionary and delta decoders.
The kernels must neither fault nor write outside their buffers, and status, values_written and the
values delivered must be those of the oracle (the Java decoders throw at the same value)."""
import numpy as np
import pytest


def junk_streams(rng, count, max_len):
    """Byte strings that look like hybrid streams often enough to get deep into the decoders: small
    header-like bytes mixed with arbitrary ones."""
    out = []
    for _ in range(count):
        n = int(rng.integers(0, max_len))
        kind = rng.integers(0, 4)
        if kind == 0:
            b = rng.integers(0, 256, n, dtype=np.uint8)
        elif kind == 1:
            b = rng.integers(0, 32, n, dtype=np.uint8)
        else:
            b = rng.integers(0, 256, n, dtype=np.uint8)
            b[rng.random(n) < 0.5] = rng.integers(0, 12)
            if kind == 3 and n:
                b[rng.integers(0, n, max(n // 16, 1))] = rng.integers(128, 256)
        out.append(b)
    return out


def test_oracle_survives_garbage(orc):
    rng = np.random.default_rng(404)
    for data in junk_streams(rng, 300, 300):
        w = int(rng.integers(0, 33))
        n = int(rng.integers(0, 3000))
        st, vals, used = orc.rle_decode(w, data, n)
        assert 0 <= st <= 5 and vals.size <= n
        st, vals, total, used = orc.delta_decode(data, 4000)
        assert 0 <= st <= 5


@pytest.mark.gpu
def test_gpu_hybrid_garbage(orc, rle_path):
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(405)
    pages, expect = [], []
    for data in junk_streams(rng, 600, 400):
        w = int(rng.integers(0, 33))
        n = int(rng.integers(0, 5000))
        flags = int(rng.choice([0, 0, pq.F_LENGTH_PREFIX, pq.F_WIDTH_PREFIX]))
        if flags == pq.F_WIDTH_PREFIX:
            ost, ovals = orc.dict_decode_ids(data, n)
            if data.size and data[0] > 32:
                continue  # widths above 32: the readers reject them in different ways, covered elsewhere
        else:
            ost, ovals, _ = orc.rle_decode(w, data, n, prefixed=bool(flags))
        pages.append(gh.Page(data, n, pq.ENC_RLE, 4, bit_width=w, flags=flags))
        expect.append((ost, ovals))
    for i, ((st, written, _, out, guard), (ost, ovals)) in enumerate(zip(gh.decode_pages(pages), expect)):
        assert st == ost and written == ovals.size, (i, st, ost, written, ovals.size, pages[i].bit_width, pages[i].flags)
        assert (out.view(np.int32)[:written] == ovals).all(), i
        assert (guard == 0xA5).all()


@pytest.mark.gpu
@pytest.mark.parametrize("elem", [4, 8])
def test_gpu_dictionary_garbage(orc, elem, dict_path):
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(406 + elem)
    dt = np.int32 if elem == 4 else np.int64
    dicts = [rng.integers(-1000, 1000, k).astype(dt) for k in (1, 7, 300, 5000)]
    pages, expect = [], []
    for data in junk_streams(rng, 400, 400):
        if data.size and data[0] > 32:
            data = data.copy()
            data[0] = rng.integers(0, 33)
        dic = dicts[int(rng.integers(0, len(dicts)))]
        n = int(rng.integers(0, 5000))
        pages.append(gh.Page(data, n, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic))
        expect.append(orc.dict_decode(data, dic, n))
    for i, ((st, written, _, out, guard), (ost, ovals)) in enumerate(zip(gh.decode_pages(pages), expect)):
        assert st == ost and written == ovals.size, (i, st, ost, written, ovals.size)
        assert (out.view(dt)[:written] == ovals).all(), i
        assert (guard == 0xA5).all()


@pytest.mark.gpu
@pytest.mark.parametrize("elem", [4, 8])
def test_gpu_delta_garbage(orc, elem):
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(407 + elem)
    dt = np.int32 if elem == 4 else np.int64
    pages, expect = [], []
    good = orc.delta_encode(rng.integers(-1000, 1000, 700).astype(dt))
    for k, data in enumerate(junk_streams(rng, 300, 300)):
        if k % 3 == 0:  # a valid page with a few bytes damaged, or cut short
            data = good.copy()
            hits = rng.integers(4, data.size, int(rng.integers(1, 4)))
            data[hits] = rng.integers(0, 256, hits.size)
            data = data[:int(rng.integers(1, data.size + 1))]
        cap = int(rng.integers(0, 3000))
        ost, ovals, total, used = orc.delta_decode(data, cap)
        pages.append(gh.Page(data, cap, pq.ENC_DELTA_BINARY_PACKED, elem))
        expect.append((ost, ovals.astype(dt) if ost == 0 else ovals, total))
    for i, ((st, written, _, out, guard), (ost, ovals, total)) in enumerate(zip(gh.decode_pages(pages), expect)):
        assert st == ost, (i, st, ost)
        if ost == 0:
            assert written == total and (out.view(dt)[:written] == ovals[:written]).all(), i
        assert (guard == 0xA5).all()


@pytest.mark.gpu
@pytest.mark.parametrize("elem", [4, 8])
def test_gpu_delta_headers_with_huge_miniblocks(orc, elem):
    """Block sizes near 2^30 with one miniblock per block: miniblock size x width / 8 does not fit 32 bits
    (2^27 x 32 = 2^32).  The reference runs off the stream in in.slice() (EOFException); a product taken
    modulo 2^32 would accept the miniblock as empty."""
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)

    def varint(v):
        out = bytearray()
        while v >= 0x80:
            out.append((v & 0x7F) | 0x80)
            v >>= 7
        out.append(v)
        return bytes(out)

    pages, expect = [], []
    for block, mini, width, total, tail in ((1 << 30, 1, 32, 100, 64), (1 << 30, 1, 64, 100, 64), (1 << 30, 2, 64, 5, 0),
                                            (1 << 29, 1, 64, 3000, 4000), (1 << 22, 1, 0, 50, 8), (1 << 16, 1, 2, 70, 1 << 14),
                                            (1 << 16, 1, 3, 70, 100), ((1 << 31) - 8, 1, 8, 10, 200)):
        data = varint(block) + varint(mini) + varint(total) + b"\x00" + b"\x02" + bytes([width] * mini) + bytes(range(256)) * (tail // 256) + bytes(tail % 256)
        data = np.frombuffer(data, np.uint8)
        ost, ovals, ototal, used = orc.delta_decode(data, 3000)
        pages.append(gh.Page(data, 3000, pq.ENC_DELTA_BINARY_PACKED, elem))
        expect.append((ost, ovals, ototal))
    dt = np.int32 if elem == 4 else np.int64
    for i, ((st, written, _, out, guard), (ost, ovals, total)) in enumerate(zip(gh.decode_pages(pages), expect)):
        if ost == 0 or st != pq.DECODE:  # (a miniblock wider than the staging window is this library's documented limit)
            assert st == ost, (i, st, ost)
        if ost == 0 and st == 0:
            assert written == total and (out.view(dt)[:written] == ovals[:written].astype(dt)).all(), i
        assert (guard == 0xA5).all()


def damaged_regular_pages(orc, rng, count, widths, sizes):
    """Pages as parquet-java writes them for non-repeating values (equal bit-packed runs), with a few
    bytes overwritten (run headers among them), cut short, or asked for more or fewer values."""
    out = []
    for k in range(count):
        w = int(widths[k % len(widths)])
        n = int(sizes[k % len(sizes)])
        vals = rng.integers(0, 1 << w, n, dtype=np.uint64).astype(np.uint32).view(np.int32)
        if k % 4 == 1:
            vals[n // 3:n // 3 + 300] = vals[0]  # an RLE run inside
        data = orc.rle_encode(w, vals).copy()
        stride = 1 + 63 * w
        for _ in range(int(rng.integers(0, 4))):
            if rng.random() < 0.6:  # a run header
                at = int(rng.integers(0, max(data.size // stride, 1))) * stride
            else:
                at = int(rng.integers(0, data.size))
            data[min(at, data.size - 1)] = rng.integers(0, 256)
        if k % 3 == 0:
            data = data[:int(rng.integers(max(data.size // 2, 1), data.size + 1))]
        want = n if k % 5 else int(n * rng.uniform(0.3, 1.4))
        out.append((w, data, want))
    return out


@pytest.mark.gpu
def test_gpu_damaged_regular_hybrid_pages(orc, rle_path):
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(511)
    cases = damaged_regular_pages(orc, rng, 60, (1, 2, 3, 5, 8, 12, 16, 20, 32), (70_000, 33_000, 140_000, 9_000, 65_536))
    pages = [gh.Page(d, n, pq.ENC_RLE, 4, bit_width=w) for w, d, n in cases]
    for i, ((w, d, n), (st, written, consumed, out, guard)) in enumerate(zip(cases, gh.decode_pages(pages))):
        ost, ovals, oused = orc.rle_decode(w, d, n)
        assert st == ost and written == ovals.size, (i, w, n, st, ost, written, ovals.size)
        assert (out.view(np.int32)[:written] == ovals).all(), (i, w)
        if ost == 0 and n:
            assert consumed == oused, (i, consumed, oused)
        assert (guard == 0xA5).all()


@pytest.mark.gpu
@pytest.mark.parametrize("elem", [4, 8])
def test_gpu_damaged_regular_dictionary_pages(orc, elem, dict_path):
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(512 + elem)
    dt = np.int32 if elem == 4 else np.int64
    cases = damaged_regular_pages(orc, rng, 40, (2, 7, 10, 12, 16), (20_000, 45_000, 70_000, 3_000))
    pages, expect = [], []
    for w, d, n in cases:
        dic = rng.integers(-(1 << 31), 1 << 31, max((1 << w) - int(rng.integers(0, 3)), 1)).astype(dt)
        data = np.concatenate([np.array([w], np.uint8), d])
        pages.append(gh.Page(data, n, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic))
        expect.append(orc.dict_decode(data, dic, n))
    for i, ((st, written, _, out, guard), (ost, ovals)) in enumerate(zip(gh.decode_pages(pages), expect)):
        assert st == ost and written == ovals.size, (i, cases[i][0], st, ost, written, ovals.size)
        assert (out.view(dt)[:written] == ovals).all(), i
        assert (guard == 0xA5).all()


@pytest.mark.gpu
@pytest.mark.parametrize("elem", [4, 8])
def test_gpu_run_heavy_dictionary_pages_damaged(orc, elem, monkeypatch):
    """Run-heavy dictionary pages of random shape (the pages the left-over launch decodes whole: ids of 1..24 bits,
    runs of random lengths, dictionaries shorter than the ids reach), as written and with a few bytes damaged, cut
    short or followed by junk: status, count and values are the oracle's, and the consumed bytes are those of the
    chunked path (PQD_WHOLE_MAX=0)."""
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(4100 + elem)
    dt = np.int32 if elem == 4 else np.int64
    pages, expect = [], []
    for k in range(240):
        w = int(rng.integers(1, 25))
        reach = 1 << w
        n_dict = int(min(reach, rng.integers(1, 70000))) if rng.random() < 0.8 else int(rng.integers(1, reach + 1) // 2 + 1)
        dic = rng.integers(-(1 << 31), 1 << 31, n_dict).astype(dt)
        n = int(rng.choice([1, 7, 8, 9, 500, 4096, 20000, 32768, int(rng.integers(1, 32769))]))
        n_runs = int(rng.integers(1, max(n // int(rng.integers(9, 120)), 1) + 1))
        ids = np.sort(rng.integers(0, reach, n_runs)).repeat(rng.integers(1, 2 * max(n // n_runs, 1) + 2, n_runs))[:n]
        ids = np.concatenate([ids, np.full(n - ids.size, ids[-1])]).astype(np.int64)
        if k % 5 == 0:  # not sorted: the runs in random order
            cut = np.flatnonzero(np.diff(ids)) + 1
            parts = np.split(ids, cut)
            ids = np.concatenate([parts[i] for i in rng.permutation(len(parts))])
        data = np.concatenate([np.array([w], np.uint8), orc.rle_encode(w, ids.astype(np.int32))])
        kind = k % 4
        if kind == 1 and data.size > 2:  # a few bytes damaged
            data = data.copy()
            hits = rng.integers(1, data.size, int(rng.integers(1, 4)))
            data[hits] = rng.integers(0, 256, hits.size)
        elif kind == 2:  # cut short
            data = data[:int(rng.integers(1, data.size + 1))]
        elif kind == 3 and k % 8 == 3:  # junk behind the stream
            data = np.concatenate([data, rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)])
        pages.append(gh.Page(data, n, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic))
        expect.append(orc.dict_decode(data, dic, n))
    results = {}
    for mode in ("0", None):
        if mode is None:
            monkeypatch.delenv("PQD_WHOLE_MAX")
        else:
            monkeypatch.setenv("PQD_WHOLE_MAX", mode)
        results[mode] = gh.decode_pages(pages)
        for i, ((st, written, _, out, guard), (ost, ovals)) in enumerate(zip(results[mode], expect)):
            assert st == ost and written == ovals.size, (mode, i, st, ost, written, ovals.size)
            assert (out.view(dt)[:written] == ovals).all(), (mode, i)
            assert (guard == 0xA5).all(), (mode, i)
    for i, (a, b) in enumerate(zip(results["0"], results[None])):
        assert a[2] == b[2] or a[0] != 0, (i, a[2], b[2])  # bytes consumed (where the page decoded)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["plain", "dict4", "dict8"])
def test_gpu_long_run_heavy_pages_damaged(orc, kind, monkeypatch):
    """Long pages of many short runs of random shape (the pages k_whole_long_pages takes: 33 000 to 600 000 values, widths
    1..24, runs of random mean length, long RLE runs and bit-packed stretches mixed in), as written and with a few bytes
    damaged, cut short, followed by junk, or asked for more or fewer values than they hold: status, count and values are
    the oracle's, and the consumed bytes are those of the planned path (PQD_WHOLE_MAX=0)."""
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng({"plain": 8100, "dict4": 8104, "dict8": 8108}[kind])
    elem = 8 if kind == "dict8" else 4
    dt = np.int64 if elem == 8 else np.int32
    pages, expect = [], []
    for k in range(36):
        w = int(rng.integers(1, 25))
        n = int(rng.choice([33_000, 70_000, 150_000, 262_144, 262_145, 600_000]))
        mean = int(rng.choice([9, 15, 40, 63]))
        lens = rng.geometric(1.0 / mean, 2 * n // mean + 64)
        vals = np.repeat(rng.integers(0, 1 << w, lens.size), lens)[:n].astype(np.int64)
        if vals.size < n:
            vals = np.concatenate([vals, np.full(n - vals.size, vals[-1])])
        if k % 3 == 0:  # an RLE run of several windows' worth of values
            a = int(rng.integers(0, n // 2))
            vals[a:a + int(rng.integers(40_000, 400_000))] = int(rng.integers(0, 1 << w))
        if k % 4 == 1:  # a stretch without repeats
            a = int(rng.integers(0, n - 3000))
            vals[a:a + 3000] = rng.integers(0, 1 << w, 3000)
        body = orc.rle_encode(w, vals.astype(np.uint32).view(np.int32))
        damage = k % 5
        if damage == 1:
            body = body.copy()
            hits = rng.integers(0, body.size, int(rng.integers(1, 4)))
            body[hits] = rng.integers(0, 256, hits.size)
        elif damage == 2:
            body = body[:int(rng.integers(body.size // 2, body.size + 1))]
        elif damage == 3:
            body = np.concatenate([body, rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)])
        want = n if k % 6 else int(n * rng.uniform(0.4, 1.2))
        if kind == "plain":
            flavour = k % 3
            if flavour == 0:
                data, flags, bw = body, 0, w
                ost, ovals, _ = orc.rle_decode(w, bytes(body), want)
            elif flavour == 1:
                data, flags, bw = np.concatenate([np.array([w], np.uint8), body]), pq.F_WIDTH_PREFIX, 0
                ost, ovals = orc.dict_decode_ids(data, want)
            else:
                data = np.concatenate([np.frombuffer(int(body.size).to_bytes(4, "little"), np.uint8), body, np.array([0x99, 0x98], np.uint8)])
                flags, bw = pq.F_LENGTH_PREFIX, w
                ost, ovals, _ = orc.rle_decode(w, bytes(data), want, prefixed=True)
            pages.append(gh.Page(data, want, pq.ENC_RLE, 4, bit_width=bw, flags=flags))
            expect.append((ost, ovals))
        else:
            reach = 1 << w
            n_dict = int(min(reach, rng.integers(1, 70000))) if rng.random() < 0.7 else int(reach // 2 + 1)
            dic = rng.integers(-(1 << 31), 1 << 31, n_dict).astype(dt)
            data = np.concatenate([np.array([w], np.uint8), body])
            pages.append(gh.Page(data, want, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic))
            expect.append(orc.dict_decode(data, dic, want))
    results = {}
    for mode in ("0", None):
        if mode is None:
            monkeypatch.delenv("PQD_WHOLE_MAX")
        else:
            monkeypatch.setenv("PQD_WHOLE_MAX", mode)
        results[mode] = gh.decode_pages(pages)
        for i, ((st, written, _, out, guard), (ost, ovals)) in enumerate(zip(results[mode], expect)):
            assert st == ost and written == ovals.size, (mode, i, st, ost, written, ovals.size)
            assert (out.view(dt)[:written] == ovals).all(), (mode, i)
            assert (guard == 0xA5).all(), (mode, i)
    for i, (a, b) in enumerate(zip(results["0"], results[None])):
        assert a[2] == b[2] or a[0] != 0, (i, a[2], b[2])  # bytes consumed (where the page decoded)


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


@pytest.mark.gpu
@pytest.mark.parametrize("elem", [4, 8])
def test_gpu_dictionary_bit_packed_runs_of_growing_length(orc, elem, dict_path):
    """Bit-packed runs at one distance from each other with a LONGER run behind them (39 groups, then 158; 8, 8, 8,
    then 100; one run that announces more groups than the stream holds): the chunk is not one of equal runs, whatever
    the distance between the first two says."""
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(5150 + elem)
    dt = np.int32 if elem == 4 else np.int64
    pages, expect = [], []
    for w, groups, dict_len, declared_more in [(1, [39, 158], 7, 0), (1, [39, 158], 7, 5524), (3, [8, 8, 8, 100], 8, 0),
                                               (4, [16, 40], 16, 0), (7, [5, 5, 64], 128, 0), (12, [4, 4, 4, 4, 63], 4096, 0),
                                               (16, [3, 30], 65536, 0), (2, [63, 63, 200], 4, 0)]:
        dic = rng.integers(-10**6, 10**6, dict_len).astype(dt)
        stream = bytearray([w])
        n = 0
        for i, g in enumerate(groups):
            ids = rng.integers(0, min(dict_len, 1 << w), g * 8)
            bits = ((ids[:, None] >> np.arange(w)) & 1).astype(np.uint8).reshape(-1)
            announced = g + (declared_more if i == len(groups) - 1 else 0)
            stream += _varint((announced << 1) | 1) + np.packbits(bits, bitorder="little").tobytes()
            n += g * 8
        data = np.frombuffer(bytes(stream), dtype=np.uint8)
        for n_page in (n, n - 3):
            pages.append(gh.Page(data, n_page, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic))
            expect.append(orc.dict_decode(data, dic, n_page))
    for i, ((st, written, _, out, guard), (ost, ovals)) in enumerate(zip(gh.decode_pages(pages), expect)):
        assert ost == 0 and st == ost and written == ovals.size, (i, st, ost, written, ovals.size)
        assert (out.view(dt)[:written] == ovals).all(), i
        assert (guard == 0xA5).all()
