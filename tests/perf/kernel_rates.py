#!/usr/bin/env python3
"""Device-resident rate of every decode kernel on its reference-shaped workload, as algorithmic
bytes (encoded bytes read + decoded bytes written) per second against the 8 TB/s HBM peak.
Every page of a batch is distinct (made by the oracle's CPU encoders from its own random values, on a
thread pool), every batch reads and writes more than the 256 MiB Infinity Cache holds, and the timing loop
puts 1 GiB of unrelated traffic between two calls (see timeit), so a launch cannot be served from cache by
the launch before it.

    python tests/perf/kernel_rates.py

It lives under tests/ because it uses the oracle's ENCODERS to synthesise pages (the oracle is test
infrastructure; nothing outside tests/, the smoke check and the benchmark's CPU baseline touches it).
"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import parquet_java_amd as pq  # noqa: E402
from oracle import oracle as orc  # noqa: E402

DEV = torch.device("cuda:0")
PEAK = 8000.0


SPOILER = None


def timeit(fn, iters=10, warm=3):
    """Mean device time of fn(): events around every call, and between the calls a read of a 512 MiB buffer
    that has nothing to do with them.  Without it a loop over one call can find its input still in the
    256 MiB Infinity Cache: nontemporal output stores do not displace it, and the dictionary gather was
    measured 25 % faster that way than when its pages come from HBM.  (The traffic in between is read-only on
    purpose: a buffer written in between leaves the caches full of dirty lines, and the timed call then pays
    for writing them back: bulk unpack read 30 % slower that way.)"""
    global SPOILER
    if SPOILER is None:
        SPOILER = torch.zeros(64 << 20, dtype=torch.int64, device=DEV)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        if not os.environ.get("KR_NO_SPOILER"):  # (development: the loop as it was, input possibly cache-resident)
            SPOILER.sum()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / iters * 1e-3


ONLY = os.environ.get("KR_ONLY", "")


POOL = ThreadPoolExecutor(int(os.environ.get("KR_THREADS", "12")))


def distinct_pages(make, n_pages):
    """make(rng) -> page bytes; one generator per page, seeded by the page number."""
    return list(POOL.map(lambda i: np.frombuffer(bytes(make(np.random.default_rng(1000 + i))), np.uint8), range(n_pages)))


def report(label, t, nv, out_bytes, alg):
    print(f"{label:66s} {t * 1e6:9.1f} us  {nv / t / 1e9:7.1f} Gval/s  {out_bytes / t / 1e9:8.1f} GB/s out  "
          f"{alg / t / 1e9:8.1f} GB/s alg  {alg / t / 1e9 / PEAK * 100:5.1f}% of peak", flush=True)


def run_pages(label, make, n_pages, n_values, encoding, elem, bit_width=0, flags=0, dictionary=None):
    if ONLY and ONLY not in label:
        return
    pages = distinct_pages(make, n_pages)
    offs = np.cumsum([0] + [p.size for p in pages])
    blob = torch.from_numpy(np.concatenate(pages + [np.zeros(64, np.uint8)])).to(DEV)
    out = torch.zeros(n_pages * n_values * elem + 64, dtype=torch.uint8, device=DEV)
    d_dict = None if dictionary is None else torch.from_numpy(np.ascontiguousarray(dictionary).view(np.uint8)).to(DEV)
    descs = [dict(data=blob.data_ptr() + int(offs[i]), out=out.data_ptr() + i * n_values * elem,
                  dict=0 if d_dict is None else d_dict.data_ptr(), data_len=pages[i].size, num_values=n_values,
                  dict_len=0 if dictionary is None else dictionary.size, encoding=encoding, elem_size=elem,
                  bit_width=bit_width, flags=flags) for i in range(n_pages)]
    d_descs = torch.from_numpy(pq.descs_to_numpy(pq.make_descs(descs))).to(DEV)
    d_res = torch.zeros(n_pages * 16, dtype=torch.uint8, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    t = timeit(lambda: pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), n_pages, encoding, elem, stream))
    status, written, _ = pq.results_from_numpy(d_res.cpu().numpy())
    assert (status == 0).all() and (written == n_values).all(), (label, status[:4], written[:4])
    nv = n_pages * n_values
    report(label, t, nv, nv * elem, int(offs[-1]) + nv * elem)


def main():
    pq.init(0)
    rng = np.random.default_rng(5)
    stream = torch.cuda.current_stream().cuda_stream
    # raw bit-packed runs (the ParquetReadRouter contract)
    n = 1 << 28
    for w, long in ((8, False), (16, False), (12, True)):
        if ONLY:
            break
        nb = n * w // 8
        src = torch.randint(0, 256, (nb,), dtype=torch.uint8, device=DEV)
        dst = torch.empty(n, dtype=torch.int64 if long else torch.int32, device=DEV)
        f = pq.lib.pqd_unpack_i64_dev if long else pq.lib.pqd_unpack_i32_dev
        t = timeit(lambda: f(w, src.data_ptr(), nb, dst.data_ptr(), n, stream))
        report("unpack " + ("i64" if long else "i32") + f" w={w}, one run of 2^28 values", t, n, n * (8 if long else 4),
               nb + n * (8 if long else 4))
        del src, dst
    # RLE / bit-packing hybrid, as parquet-java writes it (1 M values per page)
    nvals = 1_000_000
    for w in (3, 8, 16):
        run_pages(f"RLE hybrid w={w}, random values, 128 pages x 1M", lambda r, w=w: orc.rle_encode(w, r.integers(0, 1 << w, nvals).astype(np.int32)),
                  128, nvals, pq.ENC_RLE, 4, bit_width=w)
    run_pages("RLE hybrid w=8, ONE page of 1M values (latency)", lambda r: orc.rle_encode(8, r.integers(0, 256, nvals).astype(np.int32)), 1,
              nvals, pq.ENC_RLE, 4, bit_width=8)

    def runs_page(r):
        v = np.repeat(r.integers(0, 256, nvals // 30), r.geometric(1 / 40.0, nvals // 30))[:nvals].astype(np.int32)
        return orc.rle_encode(8, np.concatenate([v, np.zeros(nvals - v.size, np.int32)]))
    run_pages("RLE hybrid w=8, runs ~Geometric(40), 128 pages x 1M", runs_page, 128, nvals, pq.ENC_RLE, 4, bit_width=8)

    # levels: 20000 per page
    def prefixed(body):
        return len(body).to_bytes(4, "little") + bytes(body)

    def flat_def(r):
        return (r.random(20000) < 0.9).astype(np.int32) * 2 + (r.random(20000) < 0.5)

    def nested_def(r):
        u = r.random((3, 20000))
        return ((u[0] >= 0.1) * (1 + (u[1] >= 0.1) * (1 + (u[2] >= 0.1)))).astype(np.int32)
    run_pages("definition levels w=2 (V1 prefix), 5000 pages x 20000", lambda r: prefixed(orc.rle_encode(2, flat_def(r).astype(np.int32))), 5000,
              20000, pq.ENC_RLE, 4, bit_width=2, flags=pq.F_LENGTH_PREFIX)
    run_pages("definition levels, max 3, 10% nulls per level, 5000 x 20000", lambda r: prefixed(orc.rle_encode(2, nested_def(r))), 5000,
              20000, pq.ENC_RLE, 4, bit_width=2, flags=pq.F_LENGTH_PREFIX)
    run_pages("repetition levels, max 1, lists ~Geometric(3), 5000 x 20000",
              lambda r: prefixed(orc.rle_encode(1, (r.random(20000) >= 1 / 3.0).astype(np.int32))), 5000, 20000, pq.ENC_RLE, 4, bit_width=1,
              flags=pq.F_LENGTH_PREFIX)
    run_pages("definition levels, 1% nulls (long runs), 5000 x 20000",
              lambda r: prefixed(orc.rle_encode(1, (r.random(20000) >= 0.01).astype(np.int32))), 5000, 20000, pq.ENC_RLE, 4, bit_width=1,
              flags=pq.F_LENGTH_PREFIX)
    run_pages("definition levels V2 (no prefix), max 3, 5000 x 20000", lambda r: orc.rle_encode(2, nested_def(r)), 5000, 20000, pq.ENC_RLE, 4,
              bit_width=2)
    run_pages("definition levels, legacy BIT_PACKED, max 3, 5000 x 20000",
              lambda r: orc.pack(2, nested_def(r), msb_first=True), 5000, 20000, pq.ENC_BIT_PACKED, 4, bit_width=2)
    # dictionary pages whose dictionary fits LDS whole (4096 entries, id width 12), and the 64 Ki-entry ones
    for elem, dt in ((4, np.int32), (8, np.int64)):
        for entries in (4096, 65536):
            dic = rng.integers(-(1 << 31), 1 << 31, entries).astype(dt)
            run_pages(f"dictionary INT{elem * 8}, {entries} entries, 5000 pages x 20000",
                      lambda r, e=entries: orc.dict_encode_ids(r.integers(0, e, 20000).astype(np.int32), e), 5000, 20000,
                      pq.ENC_RLE_DICTIONARY, elem, dictionary=dic)
    # 64 Ki-entry INT64 dictionary with skewed and with run-heavy ids
    dic = rng.integers(-(1 << 62), 1 << 62, 65536).astype(np.int64)
    run_pages("dictionary INT64, 65536 entries, ids ~Zipf(1.1), 5000 x 20000",
              lambda r: orc.dict_encode_ids(np.minimum(r.zipf(1.1, 20000) - 1, 65535).astype(np.int32), 65536), 5000, 20000,
              pq.ENC_RLE_DICTIONARY, 8, dictionary=dic)

    run_pages("dictionary INT64, 65536 entries, small pages: 20000 x 5000",
              lambda r: orc.dict_encode_ids(r.integers(0, 65536, 5000).astype(np.int32), 65536), 20000, 5000,
              pq.ENC_RLE_DICTIONARY, 8, dictionary=dic)

    def sorted_runs(r):
        x = np.sort(r.integers(0, 65536, 600)).repeat(r.integers(1, 70, 600))[:20000].astype(np.int32)
        return orc.dict_encode_ids(np.concatenate([x, np.full(20000 - x.size, x[-1], np.int32)]), 65536)
    run_pages("dictionary INT64, 65536 entries, sorted runs of ids, 5000 x 20000", sorted_runs, 5000, 20000, pq.ENC_RLE_DICTIONARY, 8,
              dictionary=dic)

    def short_runs(r):  # runs of 1..20 equal ids, unsorted: a stream of ~15 KB per page, a dozen chunks of 231 segments
        x = np.repeat(r.integers(0, 65536, 2400), r.integers(1, 21, 2400))[:20000].astype(np.int32)
        return orc.dict_encode_ids(np.concatenate([x, np.full(20000 - x.size, x[-1], np.int32)]), 65536)
    run_pages("dictionary INT64, 65536 entries, short runs of ids (1..20), 5000 x 20000", short_runs, 5000, 20000, pq.ENC_RLE_DICTIONARY, 8,
              dictionary=dic)
    # the same two id streams decoded as ids (a BINARY column's readValueDictionaryId: the values are the ids)
    run_pages("dictionary ids as INT32 (w=16), sorted runs, 5000 x 20000", sorted_runs, 5000, 20000, pq.ENC_RLE, 4, flags=pq.F_WIDTH_PREFIX)
    run_pages("dictionary ids as INT32 (w=16), short runs (1..20), 5000 x 20000", short_runs, 5000, 20000, pq.ENC_RLE, 4, flags=pq.F_WIDTH_PREFIX)
    # delta binary packed timestamps
    for k in (4, 12, 20, 40):
        run_pages(f"DELTA_BINARY_PACKED INT64, deltas ~{k} bits, 5000 pages x 20000",
                  lambda r, k=k: orc.delta_encode(1_600_000_000_000_000_000 + np.cumsum(1000 + r.integers(0, 1 << k, 20000)).astype(np.int64)),
                  5000, 20000, pq.ENC_DELTA_BINARY_PACKED, 8)
    # the same 20-bit deltas, but every block's min delta takes the same number of varint bytes: all blocks of one length
    run_pages("DELTA_BINARY_PACKED INT64, deltas ~20 bits, blocks of one length, 5000 x 20000",
              lambda r: orc.delta_encode(1_600_000_000_000_000_000 + np.cumsum((1 << 20) + 1000 + r.integers(0, 1 << 20, 20000)).astype(np.int64)),
              5000, 20000, pq.ENC_DELTA_BINARY_PACKED, 8)
    run_pages("DELTA_BINARY_PACKED INT32, deltas ~8 bits, 5000 pages x 20000",
              lambda r: orc.delta_encode(np.cumsum(r.integers(0, 256, 20000)).astype(np.int32)), 5000, 20000, pq.ENC_DELTA_BINARY_PACKED, 4)
    # byte stream split, PLAIN
    for dt, k in ((np.float32, 4), (np.float64, 8)):
        run_pages(f"BYTE_STREAM_SPLIT {np.dtype(dt).name}, 5000 pages x 20000",
                  lambda r, dt=dt, k=k: orc.bss_encode(r.standard_normal(20000).astype(dt).view(np.uint8), k), 5000, 20000,
                  pq.ENC_BYTE_STREAM_SPLIT, k)
        run_pages(f"BYTE_STREAM_SPLIT {np.dtype(dt).name}, 100 pages x 1M",
                  lambda r, dt=dt, k=k: orc.bss_encode(r.standard_normal(1_000_000).astype(dt).view(np.uint8), k), 100 if k == 8 else 200,
                  1_000_000, pq.ENC_BYTE_STREAM_SPLIT, k)
    run_pages("PLAIN INT64, 5000 pages x 20000", lambda r: r.integers(-(1 << 62), 1 << 62, 20000).astype(np.int64).tobytes(), 5000, 20000,
              pq.ENC_PLAIN, 8)
    run_pages("PLAIN BOOLEAN -> bytes, 5000 pages x 20000", lambda r: r.integers(0, 256, 2500, dtype=np.uint8).tobytes(), 5000, 20000,
              pq.ENC_PLAIN_BOOLEAN, 1)
    levels_to_structure()
    byte_arrays()
    decompression()
    shard_of_mixed_row_groups()


def levels_to_structure():
    """Decoded levels of 5000 pages x 20000 -> validity bitmap + positions, null scatter of INT64 values, list offsets."""
    if ONLY and ONLY not in "levels -> structure":
        return
    n_pages, n = 5000, 20000
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=DEV)
    g.manual_seed(3)
    dl = (torch.rand((n_pages, n), device=DEV, generator=g) >= 0.1).to(torch.int32) * 3
    rl = (torch.rand((n_pages, n), device=DEV, generator=g) >= 1 / 3.0).to(torch.int32)
    rl[:, 0] = 0
    validity = torch.zeros((n_pages, n // 8), dtype=torch.uint8, device=DEV)
    index = torch.zeros((n_pages, n), dtype=torch.int32, device=DEV)
    values = torch.randint(-(1 << 62), 1 << 62, (n_pages, n), dtype=torch.int64, device=DEV, generator=g)
    spaced = torch.zeros((n_pages, n), dtype=torch.int64, device=DEV)
    offsets = torch.zeros((n_pages, n + 1), dtype=torch.int32, device=DEV)
    res = torch.zeros(n_pages * 16, dtype=torch.uint8, device=DEV)
    for with_index in (False, True):
        v = (pq.ValidityDesc * n_pages)()
        for i in range(n_pages):
            v[i].def_levels, v[i].validity = dl[i].data_ptr(), validity[i].data_ptr()
            v[i].value_index = index[i].data_ptr() if with_index else 0
            v[i].n_levels, v[i].max_def = n, 3
        d_v = torch.from_numpy(np.frombuffer(bytes(v), np.uint8).copy()).to(DEV)
        t = timeit(lambda: pq.check(pq.lib.pqd_def_levels_to_validity_dev(d_v.data_ptr(), res.data_ptr(), n_pages, stream)))
        alg = n_pages * n * 4 + n_pages * n // 8 + (n_pages * n * 4 if with_index else 0)
        report("definition levels -> validity" + (" + value positions" if with_index else "") + ", 5000 x 20000", t, n_pages * n,
               alg - n_pages * n * 4, alg)
    counts = res.cpu().numpy().view(np.uint32).reshape(-1, 4)[:, 3]
    sc = (pq.ScatterDesc * n_pages)()
    for i in range(n_pages):
        sc[i].values, sc[i].validity, sc[i].out = values[i].data_ptr(), validity[i].data_ptr(), spaced[i].data_ptr()
        sc[i].n_slots, sc[i].n_values = n, int(counts[i])
    d_sc = torch.from_numpy(np.frombuffer(bytes(sc), np.uint8).copy()).to(DEV)
    t = timeit(lambda: pq.check(pq.lib.pqd_scatter_nulls_dev(d_sc.data_ptr(), res.data_ptr(), n_pages, 8, stream)))
    report("null scatter of INT64 values (10% nulls), 5000 x 20000", t, n_pages * n, n_pages * n * 8,
           int(counts.sum()) * 8 + n_pages * n // 8 + n_pages * n * 8)
    of = (pq.OffsetsDesc * n_pages)()
    for i in range(n_pages):
        of[i].rep_levels, of[i].def_levels, of[i].offsets = rl[i].data_ptr(), dl[i].data_ptr(), offsets[i].data_ptr()
        of[i].n_levels, of[i].rows_capacity, of[i].slot_def, of[i].row_def = n, n, 2, 1
    d_of = torch.from_numpy(np.frombuffer(bytes(of), np.uint8).copy()).to(DEV)
    t = timeit(lambda: pq.check(pq.lib.pqd_rep_levels_to_offsets_dev(d_of.data_ptr(), res.data_ptr(), n_pages, stream)))
    rows = int(res.cpu().numpy().view(np.uint32).reshape(-1, 4)[:, 1].sum())
    report("repetition + definition levels -> list offsets, 5000 x 20000", t, n_pages * n, rows * 4, n_pages * n * 8 + rows * 4)


def byte_arrays():
    """BINARY dictionary values (ids -> views -> bytes) and the delta byte array encodings, 2000-5000 pages x 20000."""
    if ONLY and ONLY not in "byte arrays":
        return
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(9)
    n_pages, n = 2500, 20000
    # BINARY dictionary of 65536 values of 4..12 bytes; views as the dictionary gather delivers them
    lens = rng.integers(4, 13, 65536)
    starts = np.cumsum(lens + 4) - lens  # PLAIN BYTE_ARRAY dictionary page: 4-byte length in front of each value
    views = (starts.astype(np.uint64) | (lens.astype(np.uint64) << np.uint64(32)))
    dict_bytes = torch.randint(0, 256, (int(starts[-1] + lens[-1]) + 16,), dtype=torch.uint8, device=DEV)
    d_views_dict = torch.from_numpy(views.view(np.int64)).to(DEV)
    g = torch.Generator(device=DEV)
    g.manual_seed(4)
    ids = torch.randint(0, 65536, (n_pages, n), device=DEV, generator=g)
    page_views = d_views_dict[ids]  # distinct per page
    cap = 12 * n
    offsets = torch.zeros((n_pages, n + 1), dtype=torch.int32, device=DEV)
    out = torch.zeros((n_pages, cap), dtype=torch.uint8, device=DEV)
    b = (pq.BinaryPageDesc * n_pages)()
    for i in range(n_pages):
        b[i].views, b[i].dict_bytes, b[i].offsets, b[i].bytes = page_views[i].data_ptr(), dict_bytes.data_ptr(), offsets[i].data_ptr(), out[i].data_ptr()
        b[i].num_values, b[i].dict_bytes_len, b[i].bytes_capacity = n, dict_bytes.numel() - 16, cap
    d_b = torch.from_numpy(np.frombuffer(bytes(b), np.uint8).copy()).to(DEV)
    res = torch.zeros(n_pages * 16, dtype=torch.uint8, device=DEV)
    t = timeit(lambda: pq.check(pq.lib.pqd_binary_materialize_dev(d_b.data_ptr(), res.data_ptr(), n_pages, stream)))
    r = res.cpu().numpy().view(np.uint32).reshape(-1, 4)
    assert (r[:, 0] == 0).all()
    nbytes = int(r[:, 2].sum())
    report("BINARY dictionary values: views -> offsets + bytes (4-12 B), 2500 x 20000", t, n_pages * n, nbytes + n_pages * n * 4,
           n_pages * n * 8 + nbytes + n_pages * n * 4 + nbytes)
    del page_views, out, offsets, ids

    # delta byte arrays: sorted URL-like strings (DELTA_BYTE_ARRAY) and the same values as DELTA_LENGTH_BYTE_ARRAY
    def strings(r):
        base = b"https://example.org/" + bytes(r.integers(97, 123, 40, dtype=np.uint8))
        cut, num = r.integers(20, 61, n), r.integers(0, 1000000, n)
        return sorted(base[:c] + b"/%06d" % k for c, k in zip(cut.tolist(), num.tolist()))

    def dlba(r):
        v = strings(r)
        return np.concatenate([orc.delta_encode(np.array([len(x) for x in v], np.int32)), np.frombuffer(b"".join(v), np.uint8)])

    def dba(r):
        v = strings(r)
        lens = np.array([len(x) for x in v])
        width = int(lens.max())
        grid = np.frombuffer(b"".join(x.ljust(width, b"\0") for x in v), np.uint8).reshape(n, width)
        differ = grid[1:] != grid[:-1]
        first = np.where(differ.any(1), differ.argmax(1), np.minimum(lens[1:], lens[:-1]))
        prefixes = np.concatenate([[0], np.minimum(first, np.minimum(lens[1:], lens[:-1]))]).astype(np.int32)
        suffixes = b"".join(x[k:] for x, k in zip(v, prefixes.tolist()))
        return np.concatenate([orc.delta_encode(prefixes), orc.delta_encode((lens - prefixes).astype(np.int32)),
                               np.frombuffer(suffixes, np.uint8)])
    def plain(r):  # PLAIN BYTE_ARRAY: a 4-byte length in front of every value (BinaryPlainValuesReader)
        v = strings(r)
        return np.frombuffer(b"".join(len(x).to_bytes(4, "little") + x for x in v), np.uint8)
    for label, make, enc, n_p in (("PLAIN BYTE_ARRAY, sorted URLs ~47 B, 2560 x 20000 (320 distinct)", plain, 10, int(os.environ.get("KR_PB_PAGES", "2560"))),
                                  ("DELTA_LENGTH_BYTE_ARRAY, sorted URLs ~47 B, 2560 x 20000 (320 distinct)", dlba, 8, 2560),
                                  ("DELTA_BYTE_ARRAY, sorted URLs ~47 B, 2560 x 20000 (320 distinct)", dba, 9, 2560)):
        made = distinct_pages(make, n_p // 8)  # (building the strings in Python is slow: every page is decoded eight times,
        pages = made * 8                        # into outputs of its own; 190 MB of pages and 2.4 GB of output per launch)
        offs = np.concatenate([np.cumsum([0] + [p.size for p in made])[:-1]] * 8)  # page i lies where page i % 320 lies
        blob = torch.from_numpy(np.concatenate(made + [np.zeros(64, np.uint8)])).to(DEV)
        cap = 80 * n
        d_off = torch.zeros((n_p, n + 1), dtype=torch.int32, device=DEV)
        d_bytes = torch.zeros((n_p, cap), dtype=torch.uint8, device=DEV)
        work = torch.zeros((n_p, pq._native.delta_bytes_work(n) + 16), dtype=torch.uint8, device=DEV)
        d = (pq.DeltaBytesDesc * n_p)()
        for i in range(n_p):
            d[i].data, d[i].offsets, d[i].bytes, d[i].work = blob.data_ptr() + int(offs[i]), d_off[i].data_ptr(), d_bytes[i].data_ptr(), work[i].data_ptr()
            d[i].data_len, d[i].num_values, d[i].bytes_capacity, d[i].encoding = pages[i].size, n, cap, enc
        d_d = torch.from_numpy(np.frombuffer(bytes(d), np.uint8).copy()).to(DEV)
        res = torch.zeros(n_p * 16, dtype=torch.uint8, device=DEV)
        t = timeit(lambda: pq.check(pq.lib.pqd_delta_byte_array_dev(d_d.data_ptr(), res.data_ptr(), n_p, stream)), iters=5)
        r = res.cpu().numpy().view(np.uint32).reshape(-1, 4)
        assert (r[:, 0] == 0).all() and (r[:, 1] == n).all()
        nbytes = int(r[:, 3].sum())
        report(label, t, n_p * n, nbytes + n_p * n * 4, sum(p.size for p in pages) + nbytes + n_p * n * 4)
        del blob, d_off, d_bytes, work


def decompression():
    """SNAPPY / LZ4_RAW / ZSTD / GZIP pages of 1 MiB (parquet-java's default page size), compressed with the C libraries through pyarrow."""
    if ONLY and ONLY not in "decompress":
        return
    import pyarrow as pa
    stream = torch.cuda.current_stream().cuda_stream
    size = 1 << 20
    words = [bytes(np.random.default_rng(1).integers(97, 123, k, dtype=np.uint8)) for k in np.random.default_rng(2).integers(2, 10, 2000)]

    def text(r):
        return b" ".join(words[i] for i in r.integers(0, 2000, size // 5))[:size]

    def ints(r):
        return (r.integers(0, 1000, size // 4).astype("<i4")).tobytes()

    def doubles(r):
        return np.round(r.normal(size=size // 8), 2).astype("<f8").tobytes()
    for kind, make in (("text", text), ("int32 < 1000", ints), ("doubles, 2 decimals", doubles)):
        for codec, num in (("snappy", 1), ("lz4_raw", 7), ("zstd", 6), ("gzip", 2)):
            n_p = int(os.environ.get("KR_LZ_PAGES", "1792"))
            pages = distinct_pages(lambda r: pa.Codec(codec).compress(make(r), asbytes=True), n_p)
            offs = np.cumsum([0] + [p.size for p in pages])
            blob = torch.from_numpy(np.concatenate(pages + [np.zeros(64, np.uint8)])).to(DEV)
            out = torch.zeros((n_p, size), dtype=torch.uint8, device=DEV)
            d = (pq.DecompressDesc * n_p)()
            for i in range(n_p):
                d[i].src, d[i].dst, d[i].src_len, d[i].dst_len, d[i].codec = blob.data_ptr() + int(offs[i]), out[i].data_ptr(), pages[i].size, size, num
            d_d = torch.from_numpy(np.frombuffer(bytes(d), np.uint8).copy()).to(DEV)
            res = torch.zeros(n_p * 16, dtype=torch.uint8, device=DEV)
            t = timeit(lambda: pq.check(pq.lib.pqd_decompress_pages_dev(d_d.data_ptr(), res.data_ptr(), n_p, num, stream)), iters=3, warm=1)
            r = res.cpu().numpy().view(np.uint32).reshape(-1, 4)
            assert (r[:, 0] == 0).all() and (r[:, 1] == size).all()
            report(f"{codec.upper()} pages of 1 MiB, {kind} (ratio {n_p * size / offs[-1]:.1f}), {n_p} pages", t, n_p * size, n_p * size,
                   int(offs[-1]) + n_p * size)
            del blob, out


def shard_of_mixed_row_groups():
    """64 row groups x 20 column chunks of 250 000 values (5 each of dictionary INT32 with 12-bit ids,
    dictionary INT64 with 16-bit ids, DELTA_BINARY_PACKED INT64 and BYTE_STREAM_SPLIT DOUBLE, definition
    levels on half of them), pages of 25 000 values: the one-GPU share of the sharded workload, decoded with
    one pqd_decode_pages_dev call per encoding class."""
    label = "mixed shard: 64 row groups x 20 chunks x 250000 values"
    if ONLY and ONLY not in label:
        return
    rng = np.random.default_rng(8)
    n, pages_per_chunk, groups = 25_000, 10, 64
    d32 = rng.integers(-(1 << 31), 1 << 31, 1 << 12).astype(np.int32)
    d64 = rng.integers(-(1 << 62), 1 << 62, 1 << 16).astype(np.int64)
    u = rng.random((3, n))
    lv = ((u[0] >= 0.1) * (1 + (u[1] >= 0.1) * (1 + (u[2] >= 0.1)))).astype(np.int32)
    lv_body = orc.rle_encode(2, lv)
    def levels_page(r):
        u = r.random((3, n))
        body = orc.rle_encode(2, ((u[0] >= 0.1) * (1 + (u[1] >= 0.1) * (1 + (u[2] >= 0.1)))).astype(np.int32))
        return len(body).to_bytes(4, "little") + bytes(body)
    kinds = {  # class -> (page maker, encoding, elem, bit_width, flags, dictionary, chunks per row group)
        "dict32": (lambda r: orc.dict_encode_ids(r.integers(0, 4096, n).astype(np.int32), 4096), pq.ENC_RLE_DICTIONARY, 4, 0, 0, d32, 5),
        "dict64": (lambda r: orc.dict_encode_ids(r.integers(0, 65536, n).astype(np.int32), 65536), pq.ENC_RLE_DICTIONARY, 8, 0, 0, d64, 5),
        "delta": (lambda r: orc.delta_encode(1_600_000_000_000_000_000 + np.cumsum(1000 + r.integers(0, 1 << 12, n)).astype(np.int64)),
                  pq.ENC_DELTA_BINARY_PACKED, 8, 0, 0, None, 5),
        "bss": (lambda r: orc.bss_encode(r.standard_normal(n).view(np.uint8), 8), pq.ENC_BYTE_STREAM_SPLIT, 8, 0, 0, None, 5),
        "levels": (levels_page, pq.ENC_RLE, 4, 2, 1, None, 10),
    }
    stream = torch.cuda.current_stream().cuda_stream
    calls, total_vals, total_out, total_in, keep = [], 0, 0, 0, []
    for name, (make, enc, elem, bw, flags, dic, chunks) in kinds.items():
        n_pages = groups * chunks * pages_per_chunk
        pages = distinct_pages(make, n_pages)
        blob = torch.from_numpy(np.concatenate(pages + [np.zeros(64, np.uint8)])).to(DEV)
        offs = np.cumsum([0] + [p.size for p in pages])
        out = torch.zeros(n_pages * n * elem + 64, dtype=torch.uint8, device=DEV)
        d_dict = None if dic is None else torch.from_numpy(dic.view(np.uint8)).to(DEV)
        descs = [dict(data=blob.data_ptr() + int(offs[i]), out=out.data_ptr() + i * n * elem,
                      dict=0 if d_dict is None else d_dict.data_ptr(), data_len=pages[i].size, num_values=n,
                      dict_len=0 if dic is None else dic.size, encoding=enc, elem_size=elem, bit_width=bw, flags=flags)
                 for i in range(n_pages)]
        d_descs = torch.from_numpy(pq.descs_to_numpy(pq.make_descs(descs))).to(DEV)
        d_res = torch.zeros(n_pages * 16, dtype=torch.uint8, device=DEV)
        calls.append((d_descs, d_res, n_pages, enc, elem))
        keep += [blob, out, d_dict]
        total_vals += n_pages * n
        total_out += n_pages * n * elem
        total_in += int(offs[-1])

    def run():
        for d_descs, d_res, n_pages, enc, elem in calls:
            pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), n_pages, enc, elem, stream)
    t = timeit(run, iters=5, warm=2)
    parts = []
    for name, (d_descs, d_res, n_pages, enc, elem) in zip(kinds, calls):
        one = lambda: pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), n_pages, enc, elem, stream)
        parts.append(f"{name} {timeit(one, iters=5, warm=2) * 1e6:.0f}")
    print("    per class, alone (us): " + ", ".join(parts), flush=True)
    for d_descs, d_res, n_pages, enc, elem in calls:
        status, written, _ = pq.results_from_numpy(d_res.cpu().numpy())
        assert (status == 0).all() and (written == n).all(), (enc, status[:4])
    alg = total_in + total_out
    print(f"{label:58s} {t * 1e6:9.1f} us  {total_vals / t / 1e9:7.1f} Gval/s  {total_out / t / 1e9:8.1f} GB/s out  "
          f"{alg / t / 1e9:8.1f} GB/s alg  {alg / t / 1e9 / PEAK * 100:5.1f}% of peak  ({total_vals / 1e6:.0f} M values, "
          f"{total_out / 1e9:.2f} GB decoded)", flush=True)


if __name__ == "__main__":
    main()
