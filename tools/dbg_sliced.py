import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parquet_java_amd as pq
import bench
from explore_rle import timeit
def main():
    dev = torch.device("cuda:0"); pq.init(0)
    stream = torch.cuda.current_stream().cuda_stream
    pages = 5000
    for elem in (4, 8):
        blob, page_len, ids, dictionary, out = bench.build_column(pages, elem, 1, dev)
        d_descs, d_res = bench.make_descs(blob, page_len, dictionary, out, elem, dev)
        ms = timeit(lambda: pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), pages, pq.ENC_RLE_DICTIONARY, elem, stream))
        print(f"PQD_DBG={os.environ.get('PQD_DBG','0')} elem={elem}: {ms*1e3:8.1f} us total (plan+sliced+fallback)", flush=True)
if __name__ == "__main__":
    main()
