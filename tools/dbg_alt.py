"""INT64 dictionary call alone and alternating with the INT32 call (as bench.py's step does): per-call times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parquet_java_amd as pq
import bench


def main():
    dev = torch.device("cuda:0")
    pq.init(0)
    stream = torch.cuda.current_stream().cuda_stream
    pages = int(os.environ.get("DBG_PAGES", "5000"))
    cols = {}
    for elem in (4, 8):
        blob, page_len, ids, dictionary, out = bench.build_column(pages, elem, 1234 + elem, dev)
        d_descs, d_res = bench.make_descs(blob, page_len, dictionary, out, elem, dev)
        cols[elem] = (blob, ids, dictionary, out, d_descs, d_res)
    call = lambda e: pq.decode_pages_dev(cols[e][4].data_ptr(), cols[e][5].data_ptr(), pages, pq.ENC_RLE_DICTIONARY, e, stream)

    def timed(seq, iters=20):
        for e in seq:
            call(e)
        torch.cuda.synchronize()
        tot = {e: 0.0 for e in seq}
        evs = []
        for _ in range(iters):
            for e in seq:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                call(e)
                b.record()
                evs.append((e, a, b))
        torch.cuda.synchronize()
        for e, a, b in evs:
            tot[e] += a.elapsed_time(b)
        return {e: round(tot[e] / iters * 1e3, 1) for e in tot}

    print("INT64 alone      ", timed([8]), flush=True)
    print("INT32 alone      ", timed([4]), flush=True)
    print("alternating      ", timed([4, 8]), flush=True)
    print("INT64 alone again", timed([8]), flush=True)
    os.environ["PQD_DICT_REGULAR"] = "0"
    print("general kernel: INT64 alone", timed([8]), " alternating", timed([4, 8]), flush=True)


if __name__ == "__main__":
    main()
