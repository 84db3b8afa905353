#!/usr/bin/env python3
"""End-to-end rate of a COMPRESSED column chunk, PCIe copies included, two ways:

  device: pinned host chunk -> H2D of the compressed bytes -> pqd_chunk_walk (host) -> pqd_chunk_decompress_dev ->
          pqd_chunk_sections_dev -> pqd_decode_pages_dev -> D2H of the values
  host:   pages decompressed on the host (pyarrow's codec bindings: the same C libraries a JVM reaches through JNI;
          one thread, as a reader thread decompresses the pages of its chunk) into pinned memory -> H2D of the
          uncompressed bodies -> sections -> decode -> D2H

The chunks are written by pyarrow (an independent writer): INT64 columns of `--values` values, required, data pages
of 1 MiB, dictionary-encoded (65 536 distinct values) or PLAIN (timestamps with small steps, so that the codecs
have something to do).  Every flow's values are compared with the values written.

    python tools/e2e_compressed.py [--values 16000000] [--steps 3] [--codecs SNAPPY,LZ4_RAW,ZSTD,GZIP]
"""
import argparse
import ctypes as C
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import pyarrow as pa  # noqa: E402
import pyarrow.parquet as pap  # noqa: E402
import parquet_java_amd as pq  # noqa: E402

CODECS = {"UNCOMPRESSED": (0, "NONE", None), "SNAPPY": (1, "SNAPPY", "snappy"), "GZIP": (2, "GZIP", "gzip"), "ZSTD": (6, "ZSTD", "zstd"),
          "LZ4_RAW": (7, "LZ4", "lz4_raw")}


def write_chunk(values, codec, dictionary):
    """The bytes of the one column chunk of a file pyarrow writes for `values` (from the first page header)."""
    table = pa.table({"v": pa.array(values, pa.int64())}, schema=pa.schema([pa.field("v", pa.int64(), nullable=False)]))
    sink = io.BytesIO()
    pap.write_table(table, sink, compression=CODECS[codec][1], use_dictionary=dictionary, data_page_size=1 << 20, row_group_size=len(values),
                    data_page_version="1.0", write_statistics=False, dictionary_pagesize_limit=4 << 20)
    raw = sink.getvalue()
    cc = pap.ParquetFile(io.BytesIO(raw)).metadata.row_group(0).column(0)
    start = cc.data_page_offset if cc.dictionary_page_offset is None else min(cc.dictionary_page_offset, cc.data_page_offset)
    return raw[start:start + cc.total_compressed_size], cc.total_uncompressed_size


def pinned(nbytes):
    p = C.c_void_p()
    pq.check(pq.lib.pqd_host_alloc_pinned(C.byref(p), max(nbytes, 16)))
    return p


def as_numpy(p, nbytes, dtype=np.uint8):
    return np.frombuffer((C.c_char * nbytes).from_address(p.value), dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--values", type=int, default=16_000_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--codecs", default="SNAPPY,LZ4_RAW,ZSTD,GZIP")
    ap.add_argument("--host-threads", type=int, default=1, help="threads that decompress pages in the host flow")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    pq.init(0)
    rng = np.random.default_rng(5)
    n = args.values
    columns = {
        # ids that wander through the dictionary (neighbouring rows share values, as in sorted or clustered data)
        "dictionary": (rng.integers(0, 1 << 62, 65536)[(np.arange(n) // 50 + rng.integers(0, 16, n)) & 0xFFFF], True),
        "plain_timestamps": (1_600_000_000_000 + np.cumsum(rng.integers(0, 2000, n)), False),
    }
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(args.host_threads) if args.host_threads > 1 else None
    stream = C.c_void_p()
    pq.check(pq.lib.pqd_stream_create(C.byref(stream)))
    s = stream.value
    h_out = pinned(n * 8)
    d_val = torch.zeros(n + 64, dtype=torch.int64, device=dev)
    print(f"host flow decompresses with {args.host_threads} thread(s)")
    print(f"{n} INT64 values per chunk ({n * 8 / 1e6:.0f} MB decoded), 1 MiB pages, {args.steps} steps; rates are decoded bytes per second of wall clock")
    for cname, (values, dictionary) in columns.items():
        for codec in ["UNCOMPRESSED"] + args.codecs.split(","):
            num, _, pa_name = CODECS[codec]
            chunk, raw_size = write_chunk(values, codec, dictionary)
            h_chunk = pinned(len(chunk))
            as_numpy(h_chunk, len(chunk))[:] = np.frombuffer(chunk, np.uint8)
            d_chunk = torch.zeros(len(chunk) + 64, dtype=torch.uint8, device=dev)
            cap_pages = 4096
            table = (pq.ChunkPage * cap_pages)()
            n_pages = C.c_uint32(0)
            pq.check(pq.lib.pqd_chunk_walk(h_chunk, len(chunk), n, 0, table, cap_pages, C.byref(n_pages)))
            np_ = n_pages.value
            pages = [table[i] for i in range(np_)]
            bodies_cap = sum((p.uncompressed_size + 15) & ~15 for p in pages) + 64
            d_bodies = torch.zeros(bodies_cap, dtype=torch.uint8, device=dev)
            h_bodies = pinned(bodies_cap)
            d_cd = torch.zeros(2 * np_ * 32, dtype=torch.uint8, device=dev)
            d_cr = torch.zeros(2 * np_ * 16, dtype=torch.uint8, device=dev)
            data_idx = [i for i, p in enumerate(pages) if p.type != pq.PAGE_DICTIONARY]
            nd = len(data_idx)
            base = np.cumsum([0] + [pages[i].num_values for i in data_idx]).astype(np.uint32)
            d_base = torch.from_numpy(base.view(np.uint8).copy()).to(dev)
            d_table = torch.zeros(nd * 48, dtype=torch.uint8, device=dev)
            descs = [torch.zeros(nd * 40, dtype=torch.uint8, device=dev) for _ in range(3)]
            res = [torch.zeros(nd * 16, dtype=torch.uint8, device=dev) for _ in range(2)]
            meta = pq.ColumnMeta(2, 0, 0, 0)  # INT64, required
            h_table = pinned(nd * 48)
            torch.cuda.synchronize()

            def decode(unc_pages, bodies_ptr):
                """sections + value decode + D2H from the table of uncompressed bodies (host list of ChunkPage)."""
                dict_page = [p for p in unc_pages if p.type == pq.PAGE_DICTIONARY]
                data = [p for p in unc_pages if p.type != pq.PAGE_DICTIONARY]
                as_numpy(h_table, nd * 48)[:] = np.frombuffer(bytes((pq.ChunkPage * nd)(*data)), np.uint8)
                pq.check(pq.lib.pqd_memcpy_h2d(d_table.data_ptr(), h_table, nd * 48, s))
                d_dict, dict_len = (bodies_ptr + dict_page[0].body_offset, dict_page[0].num_values) if dict_page else (0, 0)
                outs = pq.ChunkOutputs(bodies_ptr, 0, 0, d_val.data_ptr(), d_base.data_ptr(), d_base.data_ptr(), d_dict, dict_len, 0)
                pq.check(pq.lib.pqd_chunk_sections_dev(d_table.data_ptr(), nd, C.byref(meta), C.byref(outs), descs[0].data_ptr(),
                                                       descs[1].data_ptr(), descs[2].data_ptr(), res[0].data_ptr(), s))
                # one launch per run of pages with the same encoding (a dictionary chunk may end in PLAIN pages)
                a = 0
                while a < nd:
                    b = a
                    while b < nd and data[b].encoding == data[a].encoding:
                        b += 1
                    enc = pq.ENC_PLAIN if data[a].encoding == 0 else pq.ENC_RLE_DICTIONARY
                    pq.decode_pages_dev(descs[2].data_ptr() + 40 * a, res[1].data_ptr() + 16 * a, b - a, enc, 8, s)
                    a = b
                pq.check(pq.lib.pqd_memcpy_d2h(h_out, d_val.data_ptr(), n * 8, s))
                pq.check(pq.lib.pqd_stream_sync(s))

            def device_flow():
                pq.check(pq.lib.pqd_memcpy_h2d(d_chunk.data_ptr(), h_chunk, len(chunk), s))
                walked = (pq.ChunkPage * cap_pages)()
                k = C.c_uint32(0)
                pq.check(pq.lib.pqd_chunk_walk(h_chunk, len(chunk), n, 0, walked, cap_pages, C.byref(k)))
                out_table = (pq.ChunkPage * np_)()
                pq.check(pq.lib.pqd_chunk_decompress_dev(d_chunk.data_ptr(), walked, np_, num, d_bodies.data_ptr(), bodies_cap - 64, out_table,
                                                         d_cd.data_ptr(), d_cr.data_ptr(), s))
                decode(list(out_table), d_bodies.data_ptr())

            def host_flow():
                import threading
                per_thread = threading.local()  # (a codec object is not to be shared between threads)
                bodies = as_numpy(h_bodies, bodies_cap)
                unc, at, jobs = [], 0, []
                for p in pages:
                    q = pq.ChunkPage.from_buffer_copy(bytes(p))
                    q.body_offset, q.compressed_size, q.is_compressed = at, p.uncompressed_size, 0
                    unc.append(q)
                    jobs.append((p, at))
                    at += (p.uncompressed_size + 15) & ~15

                def one(job):
                    p, to = job
                    src = chunk[p.body_offset:p.body_offset + p.compressed_size]
                    if pa_name:
                        if not hasattr(per_thread, "codec"):
                            per_thread.codec = pa.Codec(pa_name)
                        src = per_thread.codec.decompress(src, decompressed_size=p.uncompressed_size)
                    bodies[to:to + p.uncompressed_size] = np.frombuffer(src, np.uint8)
                if pool is None:
                    for job in jobs:
                        one(job)
                else:
                    list(pool.map(one, jobs, chunksize=8))
                pq.check(pq.lib.pqd_memcpy_h2d(d_bodies.data_ptr(), h_bodies, at, s))
                decode(unc, d_bodies.data_ptr())

            line = f"{cname:17s} {codec:12s} {np_:4d} pages, chunk {len(chunk) / 1e6:7.1f} MB ({raw_size / max(len(chunk), 1):4.1f}x):"
            for name, flow in (("device", device_flow), ("host", host_flow)):
                as_numpy(h_out, n * 8)[:] = 0
                flow()
                status = pq.results_from_numpy(d_cr.cpu().numpy())[0] if name == "device" else np.zeros(1)
                ok = (status == 0).all() and (pq.results_from_numpy(res[1].cpu().numpy())[0] == 0).all()
                ok = ok and np.array_equal(as_numpy(h_out, n * 8, np.int64), values)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    flow()
                dt = (time.perf_counter() - t0) / args.steps
                line += f"  {name} {dt * 1e3:8.2f} ms = {n * 8 / dt / 1e9:6.2f} GB/s{'' if ok else ' WRONG'}"
            print(line, flush=True)
            for p in (h_chunk, h_bodies, h_table):
                pq.check(pq.lib.pqd_host_free_pinned(p))
            del d_chunk, d_bodies


if __name__ == "__main__":
    main()
