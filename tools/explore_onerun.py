"""Exploration: ids-only decode when every page is ONE bit-packed run (no header walk to speak of)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parquet_java_amd as pq
import bench
from explore_rle import timeit

def main():
    dev = torch.device("cuda:0"); pq.init(0)
    stream = torch.cuda.current_stream().cuda_stream
    pages = 5000; V = bench.PAGE_VALUES
    groups = V // 8
    hdr = []
    h = (groups << 1) | 1
    while h >= 0x80:
        hdr.append((h & 0x7F) | 0x80); h >>= 7
    hdr.append(h)
    page_len = 1 + len(hdr) + V * 2
    blob = torch.randint(0, 256, (pages, page_len), dtype=torch.uint8, device=dev)
    blob[:, 0] = 16
    for i, b in enumerate(hdr): blob[:, 1 + i] = b
    out = torch.zeros((pages, V), dtype=torch.int64, device=dev)
    dictionary = torch.randint(-2**62, 2**62, (65536,), dtype=torch.int64, device=dev)
    d_res = torch.zeros(pages * 16, dtype=torch.uint8, device=dev)
    pg = [dict(data=blob.data_ptr() + i * page_len, out=out.data_ptr() + i * V * 4, data_len=page_len, num_values=V,
               encoding=pq.ENC_RLE, elem_size=4, flags=pq.F_WIDTH_PREFIX) for i in range(pages)]
    d2 = torch.from_numpy(pq.descs_to_numpy(pq.make_descs(pg))).to(dev)
    ms = timeit(lambda: pq.decode_pages_dev(d2.data_ptr(), d_res.data_ptr(), pages, pq.ENC_RLE, 4, stream))
    print(f"one-run pages, ids only: {ms*1e3:.1f} us  {(pages*page_len + pages*V*4)/ms/1e6:.1f} GB/s")
    for elem in (4, 8):
        pg = [dict(data=blob.data_ptr() + i * page_len, out=out.data_ptr() + i * V * elem, dict=dictionary.data_ptr(), dict_len=65536,
                   data_len=page_len, num_values=V, encoding=pq.ENC_RLE_DICTIONARY, elem_size=elem) for i in range(pages)]
        d3 = torch.from_numpy(pq.descs_to_numpy(pq.make_descs(pg))).to(dev)
        ms = timeit(lambda: pq.decode_pages_dev(d3.data_ptr(), d_res.data_ptr(), pages, pq.ENC_RLE_DICTIONARY, elem, stream))
        print(f"one-run pages, dict elem={elem}: {ms*1e3:.1f} us  {(pages*page_len + pages*V*elem)/ms/1e6:.1f} GB/s")
if __name__ == "__main__":
    main()
