"""Times the two dictionary-page paths on the bench workload; with a PQD_PROF build of the library
(make prof; PQD_LIB_PATH=build/prof/libpqdecode_hip.so) also prints the sliced kernel's phase cycles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parquet_java_amd as pq
import bench
from explore_rle import timeit


def main():
    dev = torch.device("cuda:0")
    pq.init(0)
    stream = torch.cuda.current_stream().cuda_stream
    pages = 5000
    bench.DICT_ENTRIES = int(os.environ.get("DBG_ENTRIES", bench.DICT_ENTRIES))  # smaller: the dictionary stays in LDS
    for elem in (4, 8):
        blob, page_len, ids, dictionary, out = bench.build_column(pages, elem, 1, dev)
        d_descs, d_res = bench.make_descs(blob, page_len, dictionary, out, elem, dev)
        call = lambda: pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), pages, pq.ENC_RLE_DICTIONARY, elem, stream)
        for path in ("generic", "sliced"):
            os.environ["PQD_DICT_PATH"] = path
            ms = timeit(call)
            print(f"elem={elem} path={path}: {ms*1e3:8.1f} us", flush=True)
        if os.environ.get("DBG_SPOIL"):  # the profiled call finds nothing of its input in the caches
            spoiler = torch.zeros(64 << 20, dtype=torch.int64, device=dev)
            spoiler.sum()
            spoiler.sum()
        os.environ["PQD_PROF_PRINT"] = "1"
        call()
        torch.cuda.synchronize()
        del os.environ["PQD_PROF_PRINT"]
        exp = dictionary[ids.long()]
        print("  sliced output correct:", bool((out == exp).all()), flush=True)


if __name__ == "__main__":
    main()
