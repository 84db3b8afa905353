"""Alternating INT32 / INT64 dictionary calls (bench.py's step) under a few settings of PQD_DICT_STAGGER."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parquet_java_amd as pq
import bench


def main():
    dev = torch.device("cuda:0")
    pq.init(0)
    stream = torch.cuda.current_stream().cuda_stream
    pages = 5000
    cols = {}
    for elem in (4, 8):
        blob, page_len, ids, dictionary, out = bench.build_column(pages, elem, 1234 + elem, dev)
        d_descs, d_res = bench.make_descs(blob, page_len, dictionary, out, elem, dev)
        cols[elem] = (blob, ids, dictionary, out, d_descs, d_res)
    call = lambda e: pq.decode_pages_dev(cols[e][4].data_ptr(), cols[e][5].data_ptr(), pages, pq.ENC_RLE_DICTIONARY, e, stream)

    def timed(iters=30):
        for e in (4, 8):
            call(e)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            call(4)
            call(8)
        b.record()
        torch.cuda.synchronize()
        return round(a.elapsed_time(b) / iters * 1e3, 1)

    for rep in range(2):
        for st in sys.argv[1:]:
            os.environ["PQD_DICT_STAGGER"] = st
            print(f"stagger {st:>3}: step {timed()} us", flush=True)


if __name__ == "__main__":
    main()
