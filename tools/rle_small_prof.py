#!/usr/bin/env python3
"""Cycles per step of k_rle_small_pages on level pages (development build: make prof, then
PQD_LIB_PATH=build/prof/libpqdecode_hip.so python tools/rle_small_prof.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import parquet_java_amd as pq  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    pq.init(0)
    dev = torch.device("cuda:0")
    n_pages, n = 5000, 20000
    pages = []
    for i in range(n_pages):
        r = np.random.default_rng(i)
        u = r.random((3, n))
        body = orc.rle_encode(2, ((u[0] >= 0.1) * (1 + (u[1] >= 0.1) * (1 + (u[2] >= 0.1)))).astype(np.int32))
        pages.append(np.frombuffer(len(body).to_bytes(4, "little") + bytes(body), np.uint8))
    offs = np.cumsum([0] + [p.size for p in pages])
    blob = torch.from_numpy(np.concatenate(pages + [np.zeros(64, np.uint8)])).to(dev)
    out = torch.zeros(n_pages * n, dtype=torch.int32, device=dev)
    descs = [dict(data=blob.data_ptr() + int(offs[i]), out=out.data_ptr() + i * n * 4, data_len=pages[i].size, num_values=n,
                  encoding=pq.ENC_RLE, elem_size=4, bit_width=2, flags=pq.F_LENGTH_PREFIX) for i in range(n_pages)]
    d_descs = torch.from_numpy(pq.descs_to_numpy(pq.make_descs(descs))).to(dev)
    d_res = torch.zeros(n_pages * 16, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), n_pages, pq.ENC_RLE, 4, stream)
    torch.cuda.synchronize()
    cyc = out.view(n_pages, n)[:, :8].cpu().numpy().astype(np.int64)
    names = ["stage", "candidates", "exits", "hop", "runs", "word counts", "values", "-"]
    print("mean cycles per step (shader clock):", {k: int(v) for k, v in zip(names, cyc.mean(0))}, "total", int(cyc[:, :7].sum(1).mean()))
    print("mean stream bytes", float(np.mean([p.size for p in pages])))


if __name__ == "__main__":
    main()
