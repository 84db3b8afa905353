#!/usr/bin/env python3
"""Phase cycles of the sliced gather (PQD_PROF build: make prof; PQD_LIB_PATH=build/prof/libpqdecode_hip.so)
on run-heavy dictionary pages: sorted runs of ids over a 64 Ki-entry INT64 dictionary."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import parquet_java_amd as pq  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    pq.init(0)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    dic = rng.integers(-(1 << 62), 1 << 62, 65536).astype(np.int64)
    pages = []
    for _ in range(4):
        x = np.sort(rng.integers(0, 65536, 600)).repeat(rng.integers(1, 70, 600))[:20000].astype(np.int32)
        x = np.concatenate([x, np.full(20000 - x.size, x[-1], np.int32)])
        pages.append(np.frombuffer(bytes(orc.dict_encode_ids(x, 65536)), np.uint8))
    n_pages, n = 5000, 20000
    blob = torch.from_numpy(np.concatenate(pages)).to(dev)
    offs = np.cumsum([0] + [p.size for p in pages])
    out = torch.zeros(n_pages * n * 8, dtype=torch.uint8, device=dev)
    d_dic = torch.from_numpy(dic.view(np.uint8)).to(dev)
    descs = [dict(data=blob.data_ptr() + int(offs[i % 4]), out=out.data_ptr() + i * n * 8, dict=d_dic.data_ptr(),
                  data_len=pages[i % 4].size, num_values=n, dict_len=65536, encoding=pq.ENC_RLE_DICTIONARY, elem_size=8)
             for i in range(n_pages)]
    d_descs = torch.from_numpy(pq.descs_to_numpy(pq.make_descs(descs))).to(dev)
    d_res = torch.zeros(n_pages * 16, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), n_pages, pq.ENC_RLE_DICTIONARY, 8, stream)
    torch.cuda.synchronize()
    os.environ["PQD_PROF_PRINT"] = "1"
    pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), n_pages, pq.ENC_RLE_DICTIONARY, 8, stream)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
