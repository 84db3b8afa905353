#!/usr/bin/env python3
"""End-to-end rate of the bench workload INCLUDING the PCIe copies: encoded pages start in pinned
host memory, decoded values end in pinned host memory (what a Java reader that keeps its pages and
results on the host would see).  Per column: H2D of the page blob, decode, D2H of the values, all on
one stream; the two columns run on two streams so copies and kernels overlap.

    python tools/e2e_bench.py [--pages 5000] [--steps 5]
"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import parquet_java_amd as pq  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=5000)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    pq.init(0)
    cols = []
    for elem in (4, 8):
        blob, page_len, ids, dictionary, out = bench.build_column(args.pages, elem, 7 + elem, dev)
        d_descs, d_res = bench.make_descs(blob, page_len, dictionary, out, elem, dev)
        nb_in, nb_out = blob.numel(), out.numel() * elem
        torch.cuda.synchronize()  # the pages were built on torch's stream
        h_in, h_out, s = C.c_void_p(), C.c_void_p(), C.c_void_p()
        pq.check(pq.lib.pqd_host_alloc_pinned(C.byref(h_in), nb_in))
        pq.check(pq.lib.pqd_host_alloc_pinned(C.byref(h_out), nb_out))
        pq.check(pq.lib.pqd_stream_create(C.byref(s)))
        pq.check(pq.lib.pqd_memcpy_d2h(h_in, blob.data_ptr(), nb_in, s))  # the pages as they would sit on the host
        pq.check(pq.lib.pqd_stream_sync(s))
        expect = dictionary[ids.long()].cpu()
        cols.append(dict(elem=elem, dictionary=dictionary, ids=ids, blob=blob, out=out, descs=d_descs, res=d_res, nb_in=nb_in, nb_out=nb_out,
                         h_in=h_in, h_out=h_out, stream=s, expect=expect))
    torch.cuda.synchronize()

    def step():
        for c in cols:
            pq.check(pq.lib.pqd_memcpy_h2d(c["blob"].data_ptr(), c["h_in"], c["nb_in"], c["stream"]))
            pq.decode_pages_dev(c["descs"].data_ptr(), c["res"].data_ptr(), args.pages, pq.ENC_RLE_DICTIONARY, c["elem"],
                                c["stream"].value)
            pq.check(pq.lib.pqd_memcpy_d2h(c["h_out"], c["out"].data_ptr(), c["nb_out"], c["stream"]))
            if os.environ.get("E2E_SERIAL"):
                pq.check(pq.lib.pqd_stream_sync(c["stream"]))
        for c in cols:
            pq.check(pq.lib.pqd_stream_sync(c["stream"]))

    step()
    for c in cols:  # the values that arrived on the host are the right ones
        dt = torch.int32 if c["elem"] == 4 else torch.int64
        got = torch.frombuffer((C.c_char * c["nb_out"]).from_address(c["h_out"].value), dtype=dt)
        bad = (got != c["expect"].view(-1)).nonzero().view(-1)
        devbad = (c["out"].cpu().view(-1) != c["expect"].view(-1)).sum().item()
        assert bad.numel() == 0, f"host result differs: elem {c['elem']}, {bad.numel()} values, first at {bad[:4].tolist()}, device-side mismatches {devbad}"
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dt = (time.perf_counter() - t0) / args.steps
    n = args.pages * bench.PAGE_VALUES
    moved = sum(c["nb_in"] + c["nb_out"] for c in cols)
    print(f"end to end (pinned host -> H2D -> decode -> D2H), {args.pages} pages x {bench.PAGE_VALUES} values x 2 columns: "
          f"{dt * 1e3:.2f} ms/step, {n * 12 / dt / 1e9:.1f} GB/s decoded, {2 * n / dt / 1e9:.2f} G values/s, "
          f"{moved / dt / 1e9:.1f} GB/s over PCIe (both directions summed)")


if __name__ == "__main__":
    main()
