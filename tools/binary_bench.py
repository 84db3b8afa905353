#!/usr/bin/env python3
"""Times pqd_binary_materialize_dev alone: 5000 pages x 20000 values, 65536-entry dictionary with
entries of 4..12 bytes (the BINARY column of the benchmark workload).  PQD_LIB_PATH selects a build."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib

pq = importlib.import_module("parquet-java_amd")

PAGES, PV, D = int(os.environ.get("BB_PAGES", 5000)), 20000, 65536


def main():
    pq.init(0)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(77)
    lo, hi = int(os.environ.get("BB_MIN", 4)), int(os.environ.get("BB_MAX", 12))
    lens = rng.integers(lo, hi + 1, D)
    starts = np.cumsum(lens + 4) - lens
    page = rng.integers(0, 256, int((lens + 4).sum()), dtype=np.uint8)
    t_views = torch.from_numpy(((lens.astype(np.uint64) << np.uint64(32)) | starts.astype(np.uint64)).view(np.int64)).to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    ids = torch.randint(0, D, (PAGES, PV), device=dev, generator=g)
    views = t_views[ids].contiguous()
    d_page = torch.from_numpy(page).to(dev)
    cap = PV * hi
    offsets = torch.zeros((PAGES, PV + 1), dtype=torch.int32, device=dev)
    out = torch.zeros((PAGES, cap), dtype=torch.uint8, device=dev)
    bd = (pq.BinaryPageDesc * PAGES)()
    for i in range(PAGES):
        bd[i].views = views.data_ptr() + i * PV * 8
        bd[i].dict_bytes = d_page.data_ptr()
        bd[i].offsets = offsets.data_ptr() + i * (PV + 1) * 4
        bd[i].bytes = out.data_ptr() + i * cap
        bd[i].num_values = PV
        bd[i].dict_bytes_len = page.size
        bd[i].bytes_capacity = cap
    descs = torch.from_numpy(np.frombuffer(bytes(bd), np.uint8).copy()).to(dev)
    res = torch.zeros(PAGES * 16, dtype=torch.uint8, device=dev)

    def run():
        pq.check(pq.lib.pqd_binary_materialize_dev(descs.data_ptr(), res.data_ptr(), PAGES, stream))
    run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    st, wr, nb = pq.results_from_numpy(res.cpu().numpy())
    total = int(nb.astype(np.int64).sum())
    alg = PAGES * PV * 8 + PAGES * (PV + 1) * 4 + total
    print(f"{os.environ.get('PQD_LIB_PATH', 'default'):40s} {ms * 1e3:8.1f} us  {PAGES * PV / ms / 1e6:7.1f} Gval/s  "
          f"{alg / ms / 1e6:7.1f} GB/s alg  ok={bool((st == 0).all() and (wr == PV).all())}")


if __name__ == "__main__":
    main()
