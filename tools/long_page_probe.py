#!/usr/bin/env python3
"""Decodes pages of a million values in short runs (the case of tests/perf/kernel_rates.py) and prints the time; with a
library built with -DWP_DEBUG the whole-page routine says where it gives a page up.

    PQD_LIB_PATH=build/vard/libpqdecode_hip.so LPP_PAGES=4 python tools/long_page_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import parquet_java_amd as pq  # noqa: E402
from oracle import oracle as orc  # noqa: E402
import gpu_harness as gpu  # noqa: E402


def main():
    n_p = int(os.environ.get("LPP_PAGES", "4"))
    nvals = int(os.environ.get("LPP_VALUES", "1000000"))
    mean = float(os.environ.get("LPP_MEAN", "40"))
    pq.init(0)
    pages = []
    for i in range(n_p):
        r = np.random.default_rng(i)
        v = np.repeat(r.integers(0, 256, nvals // 30), r.geometric(1 / mean, nvals // 30))[:nvals].astype(np.int32)
        v = np.concatenate([v, np.zeros(nvals - v.size, np.int32)])
        data = orc.rle_encode(8, v)
        pages.append((gpu.Page(data, nvals, pq.ENC_RLE, 4, bit_width=8), v))
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = gpu.decode_pages([p for p, _ in pages])
        torch.cuda.synchronize()
        print(f"{n_p} pages, stream {len(pages[0][0].data)} bytes: {1e3 * (time.perf_counter() - t0):.2f} ms (with copies)")
    if os.environ.get("LPP_PROF"):  # a library built with -DPQD_PROF leaves the cycles of the first window's steps in the first values
        cyc = np.array([out.view(np.int32)[:9] for (_, _, _, out, _) in res]).astype(np.int64)
        names = ["stage", "candidates + exits", "exits of 256 bytes", "hops", "entries", "runs", "word counts", "run values", "values"]
        print("mean cycles per step of the first window:", {k: int(c) for k, c in zip(names, cyc.mean(0))})
        return
    for (st, written, used, out, guard), (p, v) in zip(res, pages):
        assert st == 0 and written == nvals and (out.view(np.int32)[:written] == v).all()
    print("values fine")


if __name__ == "__main__":
    main()
