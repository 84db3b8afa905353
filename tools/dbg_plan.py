import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import parquet_java_amd as pq
import bench
dev = torch.device("cuda:0"); pq.init(0)
stream = torch.cuda.current_stream().cuda_stream
pages = 5000
blob, page_len, ids, dictionary, out = bench.build_column(pages, 8, 1, dev)
d_descs, d_res = bench.make_descs(blob, page_len, dictionary, out, 8, dev)
for _ in range(3):
    pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), pages, pq.ENC_RLE_DICTIONARY, 8, stream)
torch.cuda.synchronize()
ph = (C.c_float * 3)(); n = C.c_int()
pq.lib.pqd_phase_ms_mean(ph, C.byref(n))
pq.lib.pqd_set_phase_timing(1)
for _ in range(20):
    pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), pages, pq.ENC_RLE_DICTIONARY, 8, stream)
pq.lib.pqd_set_phase_timing(0)
pq.lib.pqd_phase_ms_mean(ph, C.byref(n))
print(os.environ.get("PQD_LIB_PATH", "default"), "plan %.1f us, gather %.1f us" % (ph[0] * 1e3, ph[1] * 1e3), "ok" if bool((out == dictionary[ids.long()]).all()) else "WRONG")
