"""Development: the plan kernel's time for pages of several lengths and counts (phase timing of the library)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parquet_java_amd as pq
import bench
dev = torch.device("cuda:0"); pq.init(0)
stream = torch.cuda.current_stream().cuda_stream
for pages, values in ((5000, 20000), (5000, 5000), (20000, 5000), (20000, 20000), (20000, 9000)):
    bench.PAGE_VALUES = values
    blob, page_len, ids, dictionary, out = bench.build_column(pages, 8, 1, dev)
    d_descs, d_res = bench.make_descs(blob, page_len, dictionary, out, 8, dev)
    for _ in range(3):
        pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), pages, pq.ENC_RLE_DICTIONARY, 8, stream)
    torch.cuda.synchronize()
    ph = (C.c_float * 3)(); n = C.c_int()
    pq.lib.pqd_phase_ms_mean(ph, C.byref(n))
    pq.lib.pqd_set_phase_timing(1)
    for _ in range(10):
        pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), pages, pq.ENC_RLE_DICTIONARY, 8, stream)
    pq.lib.pqd_set_phase_timing(0)
    pq.lib.pqd_phase_ms_mean(ph, C.byref(n))
    ok = bool((out == dictionary[ids.long()]).all())
    print(f"{pages} pages x {values}: plan {ph[0] * 1e3:.1f} us, gather {ph[1] * 1e3:.1f} us, left-over {ph[2] * 1e3:.1f} us", "ok" if ok else "WRONG", flush=True)
    del blob, ids, out, d_descs, d_res
