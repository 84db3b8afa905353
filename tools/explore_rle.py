"""Exploration harness (not part of the product): times variants of the dictionary-page workload."""
import sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parquet_java_amd as pq
import bench

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

def main():
    dev = torch.device("cuda:0"); pq.init(0)
    stream = torch.cuda.current_stream().cuda_stream
    pages = 5000
    n_vals = pages * bench.PAGE_VALUES
    for elem in (4, 8):
        blob, page_len, ids, dictionary, out = bench.build_column(pages, elem, 1, dev)
        for dict_len in (65536, 8192, 1024, 16):
            # restrict ids by regenerating pages with ids % dict_len: rebuild blob payload
            if dict_len != 65536:
                bench.DICT_ENTRIES_SAVE = bench.DICT_ENTRIES
                g = torch.Generator(device=dev); g.manual_seed(5)
                ids2 = torch.randint(0, dict_len, (pages, bench.PAGE_VALUES), dtype=torch.int32, device=dev, generator=g)
                # patch payload in place: recompute blob via build_column logic
                old = bench.DICT_ENTRIES
                blob2 = blob.clone()
                groups = (bench.PAGE_VALUES + 7) // 8
                padded = torch.zeros((pages, groups * 8), dtype=torch.int16, device=dev)
                padded[:, :bench.PAGE_VALUES] = ids2.to(torch.int16)
                payload = padded.view(torch.uint8)
                pos = 1; gleft = groups; goff = 0
                while gleft > 0:
                    gr = min(63, gleft)
                    blob2[:, pos + 1:pos + 1 + gr * 16] = payload[:, goff * 16:(goff + gr) * 16]
                    pos += 1 + gr * 16; goff += gr; gleft -= gr
                b = blob2
            else:
                b = blob
            d_descs, d_res = bench.make_descs(b, page_len, dictionary, out, elem, dev)
            ms = timeit(lambda: pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), pages, pq.ENC_RLE_DICTIONARY, elem, stream))
            alg = pages * page_len + n_vals * elem
            print(f"dict elem={elem} dict_len={dict_len:6d}: {ms*1e3:8.1f} us  {alg/ms/1e6:8.1f} GB/s alg  {n_vals/ms/1e6:7.1f} Gval/s", flush=True)
        # ids only (no gather): PQD_ENC_RLE + WIDTH_PREFIX -> int32
        if elem == 4:
            pg = []
            for i in range(pages):
                pg.append(dict(data=blob.data_ptr() + i * page_len, out=out.data_ptr() + i * bench.PAGE_VALUES * 4,
                               data_len=page_len, num_values=bench.PAGE_VALUES, encoding=pq.ENC_RLE, elem_size=4,
                               flags=pq.F_WIDTH_PREFIX))
            d2 = torch.from_numpy(pq.descs_to_numpy(pq.make_descs(pg))).to(dev)
            ms = timeit(lambda: pq.decode_pages_dev(d2.data_ptr(), d_res.data_ptr(), pages, pq.ENC_RLE, 4, stream))
            alg = pages * page_len + n_vals * 4
            print(f"ids only (int32 out):            {ms*1e3:8.1f} us  {alg/ms/1e6:8.1f} GB/s alg  {n_vals/ms/1e6:7.1f} Gval/s", flush=True)
            # raw unpack of one 100M-value run w=16
            raw = torch.randint(0, 256, (n_vals * 2 + 64,), dtype=torch.uint8, device=dev)
            o = out.view(-1)
            for w in (4, 8, 16, 24, 32):
                nv = n_vals if w <= 16 else n_vals // 2
                ms = timeit(lambda: pq.check(pq.lib.pqd_unpack_i32_dev(w, raw.data_ptr(), nv * w // 8, o.data_ptr(), nv, stream)))
                alg = nv * w // 8 + nv * 4
                print(f"raw unpack w={w:2d} n={nv}: {ms*1e3:8.1f} us  {alg/ms/1e6:8.1f} GB/s alg", flush=True)
        del blob, ids, dictionary, out
if __name__ == "__main__":
    main()
