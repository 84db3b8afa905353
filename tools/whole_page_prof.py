#!/usr/bin/env python3
"""Cycles per step of dict_whole_page on run-heavy dictionary pages (development build: make prof, then
PQD_LIB_PATH=build/prof/libpqdecode_hip.so python tools/whole_page_prof.py)."""
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
    n_pages, n, elem = int(os.environ.get("WPP_PAGES", "5000")), int(os.environ.get("WPP_VALUES", "20000")), 8
    dic = np.random.default_rng(5).integers(-(1 << 62), 1 << 62, 65536).astype(np.int64)
    pages = []
    for i in range(n_pages):
        r = np.random.default_rng(1000 + i)
        if os.environ.get("WPP_LONG"):  # runs of about 40 equal ids (WPP_VALUES=1000000: a long page)
            x = np.repeat(r.integers(0, 65536, n // 30), r.geometric(1 / 40.0, n // 30))[:n].astype(np.int32)
        elif os.environ.get("WPP_SHORT"):  # runs of 1..20 equal ids, unsorted
            x = np.repeat(r.integers(0, 65536, 2400), r.integers(1, 21, 2400))[:n].astype(np.int32)
        else:
            x = np.sort(r.integers(0, 65536, 600)).repeat(r.integers(1, 70, 600))[:n].astype(np.int32)
        pages.append(np.frombuffer(bytes(orc.dict_encode_ids(np.concatenate([x, np.full(n - x.size, x[-1], np.int32)]), 65536)), np.uint8))
    offs = np.cumsum([0] + [p.size for p in pages])
    blob = torch.from_numpy(np.concatenate(pages + [np.zeros(64, np.uint8)])).to(dev)
    out = torch.zeros(n_pages * n, dtype=torch.int64, device=dev)
    d_dict = torch.from_numpy(dic).to(dev)
    descs = [dict(data=blob.data_ptr() + int(offs[i]), out=out.data_ptr() + i * n * elem, dict=d_dict.data_ptr(), dict_len=dic.size,
                  data_len=pages[i].size, num_values=n, encoding=pq.ENC_RLE_DICTIONARY, elem_size=elem, bit_width=0, flags=0)
             for i in range(n_pages)]
    d_descs = torch.from_numpy(pq.descs_to_numpy(pq.make_descs(descs))).to(dev)
    d_res = torch.zeros(n_pages * 16, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), n_pages, pq.ENC_RLE_DICTIONARY, elem, stream)
    torch.cuda.synchronize()
    cyc = out.view(torch.int32).view(n_pages, 2 * n)[:, :8].cpu().numpy().astype(np.int64)
    names = ["stage", "candidates + exits", "hop", "runs", "word counts", "run values", "values", "-"]
    print("mean cycles per step:", {k: int(v) for k, v in zip(names, cyc.mean(0))}, "total", int(cyc[:, :7].sum(1).mean()))
    print("mean stream bytes", float(np.mean([p.size for p in pages])))


if __name__ == "__main__":
    main()
