"""Does bench.py's alternating step still find input in the Infinity Cache?  Times the INT32 and INT64 calls
alternating (as the step does) and with 1 GiB of unrelated traffic between the calls (events around the calls only)."""
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
    scratch = torch.zeros(512 << 20, dtype=torch.uint8, device=dev)

    def timed(flush, iters=20):
        for e in (4, 8):
            call(e)
        torch.cuda.synchronize()
        tot = {4: 0.0, 8: 0.0}
        evs = []
        for _ in range(iters):
            for e in (4, 8):
                if flush:
                    scratch.add_(1)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                call(e)
                b.record()
                evs.append((e, a, b))
        torch.cuda.synchronize()
        for e, a, b in evs:
            tot[e] += a.elapsed_time(b)
        return {e: round(tot[e] / iters * 1e3, 1) for e in tot}

    for _ in range(2):
        print("alternating          ", timed(False), flush=True)
        print("alternating + flushes", timed(True), flush=True)


if __name__ == "__main__":
    main()
