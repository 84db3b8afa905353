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


def _host_page(pq, enc, elem, data, n, out, dic=None, bit_width=0, flags=0, flat=False):
    """pqd_decode_page / pqd_decode_page_flat on host arrays; returns (status of the call, PageResult)."""
    data = np.ascontiguousarray(data, np.uint8)
    res = pq.PageResult()
    dptr = dic.ctypes.data_as(C.c_void_p) if dic is not None and dic.size else None
    dlen = 0 if dic is None else dic.size
    if flat:
        st = pq.lib.pqd_decode_page_flat(enc, elem, bit_width, flags, data.ctypes.data_as(C.c_void_p), data.size, n,
                                         dptr, dlen, out.ctypes.data_as(C.c_void_p), C.byref(res))
    else:
        d = pq.PageDesc(data=data.ctypes.data, out=out.ctypes.data, dict=0 if dptr is None else dic.ctypes.data,
                        data_len=data.size, num_values=n, dict_len=dlen, encoding=enc, elem_size=elem,
                        bit_width=bit_width, flags=flags)
        st = pq.lib.pqd_decode_page(C.byref(d), C.byref(res))
    return st, res


def test_host_entry_points_from_many_threads(orc):
    """The entry points a Java ValuesReader binds (one reader per column per thread in the
    reference: RunLengthBitPackingHybridDecoder.java:47-50) called from eight host threads at once,
    none of them passing a stream: pqd_decode_page, pqd_decode_page_flat, pqd_decode_binary_page,
    pqd_unpack_i32 and pqd_decode_pages(stream = NULL).  Every result must be the oracle's."""
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    n_threads, rounds = 8, 12
    errors = []
    barrier = threading.Barrier(n_threads)

    def work(t):
        try:
            pq.init(0)
            rng = np.random.default_rng(1000 + t)
            elem = 4 if t % 2 == 0 else 8
            dt = np.int32 if elem == 4 else np.int64
            n_dict = [300, 5000, 40000, 65536][t % 4]  # one slice, several slices, the full 16-bit range
            dic = rng.integers(-(1 << 31), 1 << 31, n_dict).astype(dt)
            bin_values = [rng.integers(0, 256, int(rng.integers(0, 14)), dtype=np.uint8).tobytes() for _ in range(200 + t)]
            bin_page = orc.binary_dict_page(bin_values)
            handle = C.c_void_p()
            pq.check(pq.lib.pqd_dict_create(pq.TYPE_BYTE_ARRAY, 0, bin_page.ctypes.data_as(C.c_void_p), bin_page.size,
                                            len(bin_values), C.byref(handle)))
            _, views = orc.binary_dict_views(bin_page, len(bin_values))
            # a device-resident batch for pqd_decode_pages(NULL)
            b_ids = rng.integers(0, n_dict, (6, 3000)).astype(np.int32)
            b_datas = [orc.dict_encode_ids(b_ids[i], n_dict) for i in range(6)]
            d_blob, d_dic = gh.to_dev(np.concatenate(b_datas)), gh.to_dev(dic.view(np.uint8))
            d_out = torch.zeros(6 * 3000 * elem, dtype=torch.uint8, device=gh.dev())
            torch.cuda.synchronize()
            barrier.wait()
            for r in range(rounds):
                n = int(rng.integers(1, 30000))
                # dictionary page through the struct entry and the flat entry in turn
                ids = rng.integers(0, n_dict, n).astype(np.int32)
                if r % 3 == 0:
                    ids[n // 3:] = np.sort(ids[n // 3:])
                page = orc.dict_encode_ids(ids, n_dict)
                out = np.full(n, -7, dt)
                st, res = _host_page(pq, pq.ENC_RLE_DICTIONARY, elem, page, n, out, dic, flat=(r % 2 == 1))
                ost, ovals = orc.dict_decode(page, dic, n)
                if st != 0 or res.status != ost or res.values_written != n or not (out == ovals).all():
                    errors.append(("dict", t, r, st, res.status, res.values_written, n))
                # a level stream with a length prefix (planned hybrid path)
                lv = (rng.integers(0, 40, n) < 3).astype(np.int32) * rng.integers(0, 4, n).astype(np.int32)
                body = orc.rle_encode(2, lv)
                sect = np.concatenate([np.frombuffer(int(body.size).to_bytes(4, "little"), np.uint8), body])
                out = np.full(n, -7, np.int32)
                st, res = _host_page(pq, pq.ENC_RLE, 4, sect, n, out, bit_width=2, flags=pq.F_LENGTH_PREFIX, flat=(r % 2 == 0))
                if st != 0 or res.status != 0 or res.values_written != n or res.bytes_consumed != sect.size or not (out == lv).all():
                    errors.append(("levels", t, r, st, res.status))
                # a delta page
                ts = np.cumsum(rng.integers(-50, 5000, n)).astype(np.int64)
                dpage = orc.delta_encode(ts)
                out = np.zeros(n, np.int64)
                st, res = _host_page(pq, pq.ENC_DELTA_BINARY_PACKED, 8, dpage, n, out, flat=True)
                if st != 0 or res.status != 0 or res.values_written != n or not (out == ts).all():
                    errors.append(("delta", t, r, st, res.status))
                # BINARY page: ids -> offsets + bytes
                bids = rng.integers(0, len(bin_values), n).astype(np.int32)
                bpage = orc.dict_encode_ids(bids, len(bin_values))
                ost, ooff, obytes = orc.binary_materialize(views[bids], bin_page, 1 << 20)
                offs = np.zeros(n + 1, np.int32)
                data = np.zeros(max(obytes.size, 1), np.uint8)
                res = pq.PageResult()
                st = pq.lib.pqd_decode_binary_page(handle, bpage.ctypes.data_as(C.c_void_p), bpage.size, n,
                                                   offs.ctypes.data_as(C.c_void_p), data.ctypes.data_as(C.c_void_p),
                                                   obytes.size, C.byref(res))
                if st != 0 or res.status != 0 or res.values_written != n or not (offs == ooff).all() or \
                        not (data[:obytes.size] == obytes).all():
                    errors.append(("binary", t, r, st, res.status))
                # a bit-packed run
                w = int(rng.integers(1, 33))
                nv = int(rng.integers(1, 4000)) * 8
                packed = rng.integers(0, 256, nv // 8 * w, dtype=np.uint8)
                out = np.zeros(nv, np.int32)
                st = pq.lib.pqd_unpack_i32(w, packed.ctypes.data_as(C.c_void_p), packed.size, out.ctypes.data_as(C.c_void_p), nv)
                ost, ovals, _ = orc.read_batch(w, packed, nv)
                if st != 0 or not (out == ovals).all():
                    errors.append(("unpack", t, r, st))
                # host batch on the NULL stream, shared by every thread
                descs, pos = [], 0
                for i, d in enumerate(b_datas):
                    descs.append(dict(data=d_blob.data_ptr() + pos, out=d_out.data_ptr() + i * 3000 * elem, dict=d_dic.data_ptr(),
                                      data_len=d.size, num_values=3000, dict_len=n_dict, encoding=pq.ENC_RLE_DICTIONARY,
                                      elem_size=elem))
                    pos += d.size
                arr = pq.make_descs(descs)
                results = (pq.PageResult * 6)()
                d_out.zero_()
                torch.cuda.synchronize()
                st = pq.lib.pqd_decode_pages(arr, results, 6, None)
                got = d_out.cpu().numpy().view(dt)
                if st != 0 or any(x.status != 0 or x.values_written != 3000 for x in results) or \
                        not (got == dic[b_ids].reshape(-1)).all():
                    errors.append(("batch", t, r, st))
            pq.check(pq.lib.pqd_dict_destroy(handle))
            pq.check(pq.lib.pqd_thread_release())
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))
            try:
                barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors[:10]


def test_threads_sharing_one_stream(orc):
    """Four host threads enqueue dictionary batches on ONE caller-owned stream: the library orders
    their launch sequences, so each thread's batch decodes as if it were alone."""
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(72)
    n_pages, n = 30, 20_000
    stream = C.c_void_p()
    pq.check(pq.lib.pqd_stream_create(C.byref(stream)))
    jobs = []
    for t in range(4):
        dic = rng.integers(-(1 << 62), 1 << 62, 50_000 + t).astype(np.int64)
        ids = rng.integers(0, dic.size, (n_pages + t, n)).astype(np.int32)
        datas = [orc.dict_encode_ids(row, dic.size) for row in ids]
        d_blob, d_dic = gh.to_dev(np.concatenate(datas)), gh.to_dev(dic.view(np.uint8))
        d_out = torch.zeros(len(datas) * n * 8, dtype=torch.uint8, device=gh.dev())
        descs, pos = [], 0
        for i, d in enumerate(datas):
            descs.append(dict(data=d_blob.data_ptr() + pos, out=d_out.data_ptr() + i * n * 8, dict=d_dic.data_ptr(),
                              data_len=d.size, num_values=n, dict_len=dic.size, encoding=pq.ENC_RLE_DICTIONARY, elem_size=8))
            pos += d.size
        d_descs = gh.to_dev(pq.descs_to_numpy(pq.make_descs(descs)))
        d_res = torch.zeros(len(datas) * 16, dtype=torch.uint8, device=gh.dev())
        jobs.append(dict(expect=dic[ids].reshape(-1), d_out=d_out, d_descs=d_descs, d_res=d_res, n_pages=len(datas), keep=(d_blob, d_dic)))
    torch.cuda.synchronize()
    errors = []

    def work(job):
        for _ in range(20):
            st = pq.lib.pqd_decode_pages_dev(job["d_descs"].data_ptr(), job["d_res"].data_ptr(), job["n_pages"],
                                             pq.ENC_RLE_DICTIONARY, 8, stream)
            if st != 0:
                errors.append(st)

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    [t.start() for t in threads]
    [t.join() for t in threads]
    pq.check(pq.lib.pqd_stream_sync(stream))
    assert not errors, errors
    for j in jobs:
        status, written, _ = pq.results_from_numpy(j["d_res"].cpu().numpy())
        assert (status == 0).all() and (written == n).all()
        assert (j["d_out"].cpu().numpy().view(np.int64) == j["expect"]).all()
    pq.check(pq.lib.pqd_stream_destroy(stream))
