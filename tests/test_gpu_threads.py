"""The library is called from several host threads at once, each on its own stream (the scratch of
the planned paths is kept per device and stream): results must not depend on it."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_host_threads_on_their_own_streams(orc):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(71)
    n_threads, n_pages, n = 4, 40, 20_000
    jobs = []
    for t in range(n_threads):
        elem = 4 if t % 2 == 0 else 8
        dt = np.int32 if elem == 4 else np.int64
        dic = rng.integers(-(1 << 31), 1 << 31, 1 << 12).astype(dt)
        ids = rng.integers(0, dic.size, (n_pages, n)).astype(np.int32)
        datas = [orc.dict_encode_ids(ids[i], dic.size) for i in range(n_pages)]
        blob = np.concatenate(datas)
        d_blob, d_dic = gh.to_dev(blob), gh.to_dev(dic.view(np.uint8))
        d_out = torch.zeros(n_pages * n * elem, dtype=torch.uint8, device=gh.dev())
        descs, pos = [], 0
        for i, d in enumerate(datas):
            descs.append(dict(data=d_blob.data_ptr() + pos, out=d_out.data_ptr() + i * n * elem, dict=d_dic.data_ptr(),
                              data_len=d.size, num_values=n, dict_len=dic.size, encoding=pq.ENC_RLE_DICTIONARY, elem_size=elem))
            pos += d.size
        d_descs = gh.to_dev(pq.descs_to_numpy(pq.make_descs(descs)))
        d_res = torch.zeros(n_pages * 16, dtype=torch.uint8, device=gh.dev())
        jobs.append(dict(elem=elem, expect=dic[ids].reshape(-1), d_out=d_out, d_descs=d_descs, d_res=d_res, keep=(d_blob, d_dic)))
    torch.cuda.synchronize()
    errors = []

    def work(job):
        try:
            stream = C.c_void_p()
            pq.check(pq.lib.pqd_stream_create(C.byref(stream)))
            for _ in range(25):
                st = pq.lib.pqd_decode_pages_dev(job["d_descs"].data_ptr(), job["d_res"].data_ptr(), n_pages,
                                                 pq.ENC_RLE_DICTIONARY, job["elem"], stream)
                if st != 0:
                    errors.append(st)
            pq.check(pq.lib.pqd_stream_sync(stream))
            pq.check(pq.lib.pqd_stream_destroy(stream))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    [t.start() for t in threads]
    [t.join() for t in threads]
    torch.cuda.synchronize()
    assert not errors, errors
    for j in jobs:
        status, written, _ = pq.results_from_numpy(j["d_res"].cpu().numpy())
        assert (status == 0).all() and (written == n).all()
        got = j["d_out"].cpu().numpy().view(j["expect"].dtype)
        assert (got == j["expect"]).all()
