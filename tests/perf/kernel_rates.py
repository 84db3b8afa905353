#!/usr/bin/env python3
"""Device-resident rate of every decode kernel on its reference-shaped workload, as algorithmic
bytes (encoded bytes read + decoded bytes written) per second against the 8 TB/s HBM peak.
Pages come from the CPU encoders of the oracle (a few distinct pages, replicated on the device).

    python tests/perf/kernel_rates.py

It lives under tests/ because it uses the oracle's ENCODERS to synthesise pages (the oracle is test
infrastructure; nothing outside tests/, the smoke check and the benchmark's CPU baseline touches it).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import parquet_java_amd as pq  # noqa: E402
from oracle import oracle as orc  # noqa: E402

DEV = torch.device("cuda:0")
PEAK = 8000.0


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


ONLY = os.environ.get("KR_ONLY", "")


def run_pages(label, distinct, copies, n_values, encoding, elem, bit_width=0, flags=0, dictionary=None):
    """distinct: list of page byte strings; the batch holds each `copies` times."""
    if ONLY and ONLY not in label:
        return
    pages = [np.frombuffer(bytes(p), np.uint8) for p in distinct] * copies
    offs, pos = [], 0
    for p in pages:
        offs.append(pos)
        pos += p.size
    blob = torch.from_numpy(np.concatenate(pages)).to(DEV)
    out = torch.zeros(len(pages) * n_values * elem + 64, dtype=torch.uint8, device=DEV)
    d_dict = None if dictionary is None else torch.from_numpy(np.ascontiguousarray(dictionary).view(np.uint8)).to(DEV)
    descs = [dict(data=blob.data_ptr() + o, out=out.data_ptr() + i * n_values * elem,
                  dict=0 if d_dict is None else d_dict.data_ptr(), data_len=p.size, num_values=n_values,
                  dict_len=0 if dictionary is None else dictionary.size, encoding=encoding, elem_size=elem,
                  bit_width=bit_width, flags=flags) for i, (p, o) in enumerate(zip(pages, offs))]
    d_descs = torch.from_numpy(pq.descs_to_numpy(pq.make_descs(descs))).to(DEV)
    d_res = torch.zeros(len(pages) * 16, dtype=torch.uint8, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    t = timeit(lambda: pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), len(pages), encoding, elem, stream))
    status, written, _ = pq.results_from_numpy(d_res.cpu().numpy())
    assert (status == 0).all() and (written == n_values).all(), (label, status[:4], written[:4])
    nv = len(pages) * n_values
    alg = blob.numel() + nv * elem
    print(f"{label:58s} {t * 1e6:9.1f} us  {nv / t / 1e9:7.1f} Gval/s  {nv * elem / t / 1e9:8.1f} GB/s out  "
          f"{alg / t / 1e9:8.1f} GB/s alg  {alg / t / 1e9 / PEAK * 100:5.1f}% of peak", flush=True)


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
        alg = nb + n * (8 if long else 4)
        print(f"{'unpack ' + ('i64' if long else 'i32') + f' w={w}, one run of 2^28 values':58s} {t * 1e6:9.1f} us  "
              f"{n / t / 1e9:7.1f} Gval/s  {n * (8 if long else 4) / t / 1e9:8.1f} GB/s out  {alg / t / 1e9:8.1f} GB/s alg  "
              f"{alg / t / 1e9 / PEAK * 100:5.1f}% of peak", flush=True)
        del src, dst
    # RLE / bit-packing hybrid, as parquet-java writes it (1 M values per page)
    nvals = 1_000_000
    for w in (3, 8, 16):
        ps = [orc.rle_encode(w, rng.integers(0, 1 << w, nvals).astype(np.int32)) for _ in range(2)]
        run_pages(f"RLE hybrid w={w}, random values, 128 pages x 1M", ps, 64, nvals, pq.ENC_RLE, 4, bit_width=w)
    ps = [orc.rle_encode(8, rng.integers(0, 256, nvals).astype(np.int32))]
    run_pages("RLE hybrid w=8, ONE page of 1M values (latency)", ps, 1, nvals, pq.ENC_RLE, 4, bit_width=8)
    runs = np.repeat(rng.integers(0, 256, nvals // 30), rng.geometric(1 / 40.0, nvals // 30))[:nvals].astype(np.int32)
    run_pages("RLE hybrid w=8, runs ~Geometric(40), 128 pages x 1M", [orc.rle_encode(8, runs)], 128, nvals, pq.ENC_RLE, 4, bit_width=8)
    # levels: 20000 per page, w=2, V1 length prefix
    lv = (rng.random(20000) < 0.9).astype(np.int32) * 2 + (rng.random(20000) < 0.5)
    body = orc.rle_encode(2, lv.astype(np.int32))
    page = len(body).to_bytes(4, "little") + bytes(body)
    run_pages("definition levels w=2 (V1 prefix), 5000 pages x 20000", [page], 5000, 20000, pq.ENC_RLE, 4, bit_width=2,
              flags=pq.F_LENGTH_PREFIX if hasattr(pq, "F_LENGTH_PREFIX") else 1)
    # nested levels of a 3-deep optional schema with 10 % nulls per level, lists of ~Geometric(3) elements
    u = rng.random((3, 20000))
    dl = (u[0] >= 0.1).astype(np.int32) * (1 + (u[1] >= 0.1) * (1 + (u[2] >= 0.1)))
    body = orc.rle_encode(2, dl.astype(np.int32))
    run_pages("definition levels, max 3, 10% nulls per level, 5000 x 20000", [len(body).to_bytes(4, "little") + bytes(body)], 5000,
              20000, pq.ENC_RLE, 4, bit_width=2, flags=1)
    rl = (rng.random(20000) >= 1 / 3.0).astype(np.int32)
    body = orc.rle_encode(1, rl)
    run_pages("repetition levels, max 1, lists ~Geometric(3), 5000 x 20000", [len(body).to_bytes(4, "little") + bytes(body)], 5000,
              20000, pq.ENC_RLE, 4, bit_width=1, flags=1)
    body = orc.rle_encode(2, dl.astype(np.int32))
    run_pages("definition levels V2 (no prefix), same data, 5000 x 20000", [bytes(body)], 5000, 20000, pq.ENC_RLE, 4, bit_width=2)
    packed = orc.pack(2, np.concatenate([dl, np.zeros((-dl.size) % 8, np.int32)]).astype(np.int32), msb_first=True)
    run_pages("definition levels, legacy BIT_PACKED, same data, 5000 x 20000", [bytes(packed)], 5000, 20000, pq.ENC_BIT_PACKED, 4,
              bit_width=2)
    # dictionary pages whose dictionary fits LDS whole (4096 entries, id width 12), and the 64 Ki-entry ones
    for elem, dt in ((4, np.int32), (8, np.int64)):
        for entries in (4096, 65536):
            dic = rng.integers(-(1 << 31), 1 << 31, entries).astype(dt)
            ps = [orc.dict_encode_ids(rng.integers(0, entries, 20000).astype(np.int32), entries) for _ in range(4)]
            run_pages(f"dictionary INT{elem * 8}, {entries} entries, 5000 pages x 20000", ps, 1250, 20000,
                      pq.ENC_RLE_DICTIONARY, elem, dictionary=dic)
    # 64 Ki-entry INT64 dictionary with skewed and with run-heavy ids
    dic = rng.integers(-(1 << 62), 1 << 62, 65536).astype(np.int64)
    zipf = np.minimum(rng.zipf(1.1, (4, 20000)) - 1, 65535).astype(np.int32)
    run_pages("dictionary INT64, 65536 entries, ids ~Zipf(1.1), 5000 x 20000", [orc.dict_encode_ids(z, 65536) for z in zipf], 1250,
              20000, pq.ENC_RLE_DICTIONARY, 8, dictionary=dic)
    srt = [np.sort(rng.integers(0, 65536, 600)).repeat(rng.integers(1, 70, 600))[:20000].astype(np.int32) for _ in range(4)]
    srt = [np.concatenate([x, np.full(20000 - x.size, x[-1], np.int32)]) for x in srt]
    run_pages("dictionary INT64, 65536 entries, sorted runs of ids, 5000 x 20000", [orc.dict_encode_ids(x, 65536) for x in srt], 1250,
              20000, pq.ENC_RLE_DICTIONARY, 8, dictionary=dic)
    # delta binary packed timestamps
    for k in (4, 12, 20, 40):
        ts = 1_600_000_000_000_000_000 + np.cumsum(1000 + rng.integers(0, 1 << k, 20000)).astype(np.int64)
        ps = [orc.delta_encode(ts), orc.delta_encode(ts[::-1].copy())][:1]
        run_pages(f"DELTA_BINARY_PACKED INT64, deltas ~{k} bits, 5000 pages x 20000", ps, 5000, 20000,
                  pq.ENC_DELTA_BINARY_PACKED, 8)
    # the same 20-bit deltas, but every block's min delta takes the same number of varint bytes: all blocks of one length
    ts = 1_600_000_000_000_000_000 + np.cumsum((1 << 20) + 1000 + rng.integers(0, 1 << 20, 20000)).astype(np.int64)
    run_pages("DELTA_BINARY_PACKED INT64, deltas ~20 bits, blocks of one length, 5000 x 20000", [orc.delta_encode(ts)], 5000, 20000,
              pq.ENC_DELTA_BINARY_PACKED, 8)
    # byte stream split
    for dt, k in ((np.float32, 4), (np.float64, 8)):
        raw = rng.standard_normal(20000).astype(dt)
        run_pages(f"BYTE_STREAM_SPLIT {np.dtype(dt).name}, 5000 pages x 20000", [orc.bss_encode(raw.view(np.uint8), k)], 5000,
                  20000, pq.ENC_BYTE_STREAM_SPLIT, k)
        raw = rng.standard_normal(1_000_000).astype(dt)
        run_pages(f"BYTE_STREAM_SPLIT {np.dtype(dt).name}, 100 pages x 1M", [orc.bss_encode(raw.view(np.uint8), k)], 100,
                  1_000_000, pq.ENC_BYTE_STREAM_SPLIT, k)
    shard_of_mixed_row_groups()


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
    kinds = {  # class -> (distinct pages, encoding, elem, bit_width, flags, dictionary, pages in the shard)
        "dict32": ([orc.dict_encode_ids(rng.integers(0, 4096, n).astype(np.int32), 4096) for _ in range(4)], pq.ENC_RLE_DICTIONARY, 4, 0, 0, d32, 5),
        "dict64": ([orc.dict_encode_ids(rng.integers(0, 65536, n).astype(np.int32), 65536) for _ in range(4)], pq.ENC_RLE_DICTIONARY, 8, 0, 0, d64, 5),
        "delta": ([orc.delta_encode(1_600_000_000_000_000_000 + np.cumsum(1000 + rng.integers(0, 1 << 12, n)).astype(np.int64))], pq.ENC_DELTA_BINARY_PACKED, 8, 0, 0, None, 5),
        "bss": ([orc.bss_encode(rng.standard_normal(n).view(np.uint8), 8)], pq.ENC_BYTE_STREAM_SPLIT, 8, 0, 0, None, 5),
        "levels": ([np.concatenate([np.frombuffer(len(lv_body).to_bytes(4, "little"), np.uint8), lv_body])], pq.ENC_RLE, 4, 2, 1, None, 10),
    }
    stream = torch.cuda.current_stream().cuda_stream
    calls, total_vals, total_out, total_in, keep = [], 0, 0, 0, []
    for name, (distinct, enc, elem, bw, flags, dic, chunks) in kinds.items():
        n_pages = groups * chunks * pages_per_chunk
        pages = [np.frombuffer(bytes(p), np.uint8) for p in distinct]
        blob = torch.from_numpy(np.concatenate(pages)).to(DEV)
        offs = np.cumsum([0] + [p.size for p in pages])
        out = torch.zeros(n_pages * n * elem + 64, dtype=torch.uint8, device=DEV)
        d_dict = None if dic is None else torch.from_numpy(dic.view(np.uint8)).to(DEV)
        descs = [dict(data=blob.data_ptr() + int(offs[i % len(pages)]), out=out.data_ptr() + i * n * elem,
                      dict=0 if d_dict is None else d_dict.data_ptr(), data_len=pages[i % len(pages)].size, num_values=n,
                      dict_len=0 if dic is None else dic.size, encoding=enc, elem_size=elem, bit_width=bw, flags=flags)
                 for i in range(n_pages)]
        d_descs = torch.from_numpy(pq.descs_to_numpy(pq.make_descs(descs))).to(DEV)
        d_res = torch.zeros(n_pages * 16, dtype=torch.uint8, device=DEV)
        calls.append((d_descs, d_res, n_pages, enc, elem))
        keep += [blob, out, d_dict]
        total_vals += n_pages * n
        total_out += n_pages * n * elem
        total_in += sum(pages[i % len(pages)].size for i in range(n_pages))

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
