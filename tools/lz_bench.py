#!/usr/bin/env python3
"""Times pqd_decompress_pages_dev on pages of 1 MiB (text / small ints / rounded doubles compressed by the C libraries
through pyarrow) and compares every page with its original.

    LZB_PAGES=256 LZB_CODECS=snappy,lz4_raw LZB_KINDS=text python tools/lz_bench.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import pyarrow as pa  # noqa: E402
import parquet_java_amd as pq  # noqa: E402

NUM = {"snappy": 1, "gzip": 2, "zstd": 6, "lz4_raw": 7}


def main():
    n_p = int(os.environ.get("LZB_PAGES", "256"))
    distinct = int(os.environ.get("LZB_DISTINCT", "16"))
    size = int(os.environ.get("LZB_SIZE", str(1 << 20)))
    dev = torch.device("cuda:0")
    pq.init(0)
    stream = torch.cuda.current_stream().cuda_stream
    words = [bytes(np.random.default_rng(1).integers(97, 123, k, dtype=np.uint8)) for k in np.random.default_rng(2).integers(2, 10, 2000)]
    makers = {
        "text": lambda r: b" ".join(words[i] for i in r.integers(0, 2000, size // 5))[:size],
        "ints": lambda r: (r.integers(0, 1000, size // 4).astype("<i4")).tobytes(),
        "doubles": lambda r: np.round(r.normal(size=size // 8), 2).astype("<f8").tobytes(),
        "random": lambda r: r.integers(0, 256, size, dtype=np.uint8).tobytes(),
        "zeros": lambda r: bytes(size),
        "ids": lambda r: ((np.arange(size // 2) // 50 + r.integers(0, 16, size // 2)) & 0xFFFF).astype("<u2").tobytes(),
    }
    for kind in os.environ.get("LZB_KINDS", "text,ints,doubles").split(","):
        raws = [makers[kind](np.random.default_rng(100 + i)) for i in range(distinct)]
        for codec in os.environ.get("LZB_CODECS", "snappy,lz4_raw").split(","):
            comp = [np.frombuffer(pa.Codec(codec).compress(r, asbytes=True), np.uint8) for r in raws]
            pages = [comp[i % distinct] for i in range(n_p)]
            offs = np.cumsum([0] + [p.size for p in pages])
            blob = torch.from_numpy(np.concatenate(pages + [np.zeros(64, np.uint8)])).to(dev)
            out = torch.zeros((n_p, size), dtype=torch.uint8, device=dev)
            d = (pq.DecompressDesc * n_p)()
            for i in range(n_p):
                d[i].src, d[i].dst, d[i].src_len, d[i].dst_len, d[i].codec = blob.data_ptr() + int(offs[i]), out[i].data_ptr(), pages[i].size, size, NUM[codec]
            d_d = torch.from_numpy(np.frombuffer(bytes(d), np.uint8).copy()).to(dev)
            res = torch.zeros(n_p * 16, dtype=torch.uint8, device=dev)
            run = lambda: pq.check(pq.lib.pqd_decompress_pages_dev(d_d.data_ptr(), res.data_ptr(), n_p, NUM[codec], stream))  # noqa: E731
            run()
            torch.cuda.synchronize()
            r = res.cpu().numpy().view(np.uint32).reshape(-1, 4)
            ok = (r[:, 0] == 0).all() and (r[:, 1] == size).all()
            for i in range(min(n_p, distinct)):
                ok = ok and bytes(out[i].cpu().numpy()) == raws[i]
            if os.environ.get("LZB_PROF"):
                c = out[0][:64].cpu().numpy().view(np.uint64)
                print("   cycles/1000 of page 0: stage %d, places %d, chain %d, elements %d, masks %d, classify %d, resolve %d, store %d"
                      % tuple(int(x) // 1000 for x in c))
                if os.environ.get("LZB_PROF") == "raw":
                    print("   raw counters:", [int(x) for x in c])
                ok = True
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            t = min(ts)
            print(f"{codec:8s} {kind:8s} {n_p:5d} pages (ratio {size / comp[0].size:4.1f}): {t * 1e3:8.2f} ms  {n_p * size / t / 1e9:7.1f} GB/s out  "
                  f"{size / t / 1e6 if n_p == 1 else n_p * size / t / 1e6 / n_p:7.1f} MB/s per page in flight{'' if ok else '  WRONG'}", flush=True)
            del blob, out


if __name__ == "__main__":
    main()
