#!/usr/bin/env python3
"""Times pqd_decode_binary_pages_dev alone: 5000 pages x 20000 values, 65536-entry dictionary with entries of
4..12 bytes (the BINARY column of the benchmark workload), ids as parquet-java writes them.  PQD_LIB_PATH selects
a build; BB_MIN / BB_MAX the entry lengths; BB_SORTED=1 run-heavy ids."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import parquet_java_amd as pq  # noqa: E402
import synth_pages as sp  # noqa: E402

PAGES, PV, D = int(os.environ.get("BB_PAGES", 5000)), 20000, 65536


def main():
    pq.init(0)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(77)
    lo, hi = int(os.environ.get("BB_MIN", 4)), int(os.environ.get("BB_MAX", 12))
    lens = rng.integers(lo, hi + 1, D)
    body = rng.integers(0, 256, int(lens.sum()), dtype=np.uint8)
    page = np.zeros(int(lens.sum()) + 4 * D, np.uint8)
    starts = np.cumsum(lens + 4) - lens
    for k in range(4):
        page[starts - 4 + k] = (lens >> (8 * k)) & 0xFF
    page[np.repeat(starts - (np.cumsum(lens) - lens), lens) + np.arange(body.size)] = body
    handle = C.c_void_p()
    pq.check(pq.lib.pqd_dict_create(pq.TYPE_BYTE_ARRAY, 0, page.ctypes.data_as(C.c_void_p), page.size, D, C.byref(handle)))
    info = pq.DictInfo()
    pq.check(pq.lib.pqd_dict_get(handle, C.byref(info)))
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    ids = torch.randint(0, D, (PAGES, PV), device=dev, generator=g, dtype=torch.int32)
    if os.environ.get("BB_SORTED"):
        ids = torch.sort(ids // 64 * 64, dim=1).values
    blob, offs, w = sp.dictionary_id_pages(ids, D)
    blob = torch.cat([blob, torch.zeros(64, dtype=torch.uint8, device=dev)])
    o = offs.tolist()
    cap = PV * hi
    offsets = torch.zeros((PAGES, PV + 1), dtype=torch.int32, device=dev)
    out = torch.zeros((PAGES, cap), dtype=torch.uint8, device=dev)
    pd = pq.make_descs([dict(data=blob.data_ptr() + o[i], data_len=o[i + 1] - o[i], num_values=PV, dict_len=D,
                             encoding=pq.ENC_RLE_DICTIONARY, elem_size=8) for i in range(PAGES)])
    bd = (pq.BinaryPageDesc * PAGES)()
    for i in range(PAGES):
        bd[i].dict_bytes = info.bytes
        bd[i].offsets = offsets.data_ptr() + i * (PV + 1) * 4
        bd[i].bytes = out.data_ptr() + i * cap
        bd[i].dict_bytes_len = info.bytes_len
        bd[i].bytes_capacity = cap
    d_pd = torch.from_numpy(pq.descs_to_numpy(pd)).to(dev)
    d_bd = torch.from_numpy(np.frombuffer(bytes(bd), np.uint8).copy()).to(dev)
    pres = torch.zeros(PAGES * 16, dtype=torch.uint8, device=dev)
    bres = torch.zeros(PAGES * 16, dtype=torch.uint8, device=dev)
    spoiler = torch.zeros(64 << 20, dtype=torch.int64, device=dev)

    def run():
        pq.check(pq.lib.pqd_decode_binary_pages_dev(handle, d_pd.data_ptr(), d_bd.data_ptr(), pres.data_ptr(), bres.data_ptr(), PAGES, stream))
    run()
    torch.cuda.synchronize()
    total = 0.0
    for _ in range(8):
        spoiler.sum()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run()
        b.record()
        torch.cuda.synchronize()
        total += a.elapsed_time(b)
    ms = total / 8
    if os.environ.get("BB_STAMPS"):
        t = offsets[:, :10].cpu().numpy().astype(np.int64)
        names = ["stage", "scan", "ids", "slots", "prefix+stop", "offsets", "ORs", "flush", "-", "loop"]
        print("cycles per page (mean over pages): " + ", ".join(f"{n} {int(v)}" for n, v in zip(names, t.mean(0))), "total", int(t.sum(1).mean()))
    st, wr, nb = pq.results_from_numpy(bres.cpu().numpy())
    tot = int(nb.astype(np.int64).sum())
    alg = int(offs[-1]) + PAGES * (PV + 1) * 4 + tot
    print(f"{os.environ.get('PQD_LIB_PATH', 'default'):40s} {ms:8.4f} ms  {alg / ms / 1e6:8.1f} GB/s alg  {alg / ms / 1e6 / 80:5.1f}% of peak  "
          f"ok={bool((st == 0).all() and (wr == PV).all())}", flush=True)


if __name__ == "__main__":
    main()
