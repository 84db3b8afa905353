#!/usr/bin/env python3
"""Device-resident rate of every decode kernel on its reference-shaped workload, as algorithmic
bytes (encoded bytes read + decoded bytes written) per second against the 8 TB/s HBM peak.
Pages come from the CPU encoders of the oracle (a few distinct pages, replicated on the device).

    python tools/kernel_rates.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
    runs = np.repeat(rng.integers(0, 256, nvals // 40 + 1), rng.geometric(1 / 40.0, nvals // 40 + 1))[:nvals].astype(np.int32)
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
    # dictionary pages whose dictionary fits LDS whole (4096 entries, id width 12), and the 64 Ki-entry ones
    for elem, dt in ((4, np.int32), (8, np.int64)):
        for entries in (4096, 65536):
            dic = rng.integers(-(1 << 31), 1 << 31, entries).astype(dt)
            ps = [orc.dict_encode_ids(rng.integers(0, entries, 20000).astype(np.int32), entries) for _ in range(4)]
            run_pages(f"dictionary INT{elem * 8}, {entries} entries, 5000 pages x 20000", ps, 1250, 20000,
                      pq.ENC_RLE_DICTIONARY, elem, dictionary=dic)
    # delta binary packed timestamps
    for k in (12, 40):
        ts = 1_600_000_000_000_000_000 + np.cumsum(1000 + rng.integers(0, 1 << k, 20000)).astype(np.int64)
        ps = [orc.delta_encode(ts), orc.delta_encode(ts[::-1].copy())][:1]
        run_pages(f"DELTA_BINARY_PACKED INT64, deltas ~{k} bits, 5000 pages x 20000", ps, 5000, 20000,
                  pq.ENC_DELTA_BINARY_PACKED, 8)
    # byte stream split
    for dt, k in ((np.float32, 4), (np.float64, 8)):
        raw = rng.standard_normal(20000).astype(dt)
        run_pages(f"BYTE_STREAM_SPLIT {np.dtype(dt).name}, 5000 pages x 20000", [orc.bss_encode(raw.view(np.uint8), k)], 5000,
                  20000, pq.ENC_BYTE_STREAM_SPLIT, k)


if __name__ == "__main__":
    main()
