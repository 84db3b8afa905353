"""A row group of mixed-encoding column chunks decoded with ONE host-side batch call
(pqd_decode_pages: host descriptors, device data): the single-GPU slice of the sharded workload
(dictionary INT32 with 12-bit ids, dictionary INT64 with 16-bit ids, DELTA_BINARY_PACKED INT64,
BYTE_STREAM_SPLIT DOUBLE, definition levels), every page checked against the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_mixed_row_group_in_one_call(orc):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(2024)
    n = 25_000  # values per page (250 000-value chunks are ten of these)
    pages, expect = [], []  # (data, num_values, encoding, elem, bit_width, flags, dictionary) / expected ndarray

    d32 = rng.integers(-(1 << 31), 1 << 31, 1 << 12).astype(np.int32)
    d64 = rng.integers(-(1 << 62), 1 << 62, 1 << 16).astype(np.int64)
    for _ in range(3):
        ids = rng.integers(0, d32.size, n).astype(np.int32)
        pages.append((orc.dict_encode_ids(ids, d32.size), n, pq.ENC_RLE_DICTIONARY, 4, 0, 0, d32))
        expect.append(d32[ids])
        ids = rng.integers(0, d64.size, n).astype(np.int32)
        ids[5000:9000] = 77
        pages.append((orc.dict_encode_ids(ids, d64.size), n, pq.ENC_RLE_DICTIONARY, 8, 0, 0, d64))
        expect.append(d64[ids])
        ts = 1_600_000_000_000_000_000 + np.cumsum(1000 + rng.integers(0, 1 << 12, n)).astype(np.int64)
        pages.append((orc.delta_encode(ts), n, pq.ENC_DELTA_BINARY_PACKED, 8, 0, 0, None))
        expect.append(ts)
        dbl = rng.standard_normal(n)
        pages.append((orc.bss_encode(dbl.view(np.uint8), 8), n, pq.ENC_BYTE_STREAM_SPLIT, 8, 0, 0, None))
        expect.append(dbl.view(np.int64))
        u = rng.random((3, n))
        lv = ((u[0] >= 0.1) * (1 + (u[1] >= 0.1) * (1 + (u[2] >= 0.1)))).astype(np.int32)
        body = orc.rle_encode(2, lv)
        pages.append((np.concatenate([np.frombuffer(len(body).to_bytes(4, "little"), np.uint8), body]), n, pq.ENC_RLE, 4, 2,
                      pq.F_LENGTH_PREFIX, None))
        expect.append(lv)
    # a page of an encoding the library does not know: reported for that page only
    pages.append((np.zeros(16, np.uint8), 4, 99, 4, 0, 0, None))
    expect.append(None)

    blob = np.concatenate([np.ascontiguousarray(p[0], np.uint8) for p in pages])
    d_blob = gh.to_dev(blob)
    d_d32, d_d64 = gh.to_dev(d32.view(np.uint8)), gh.to_dev(d64.view(np.uint8))
    out_bytes = sum(p[1] * p[3] for p in pages)
    d_out = torch.zeros(out_bytes + 64, dtype=torch.uint8, device=gh.dev())
    descs = (pq.PageDesc * len(pages))()
    pos = opos = 0
    offs = []
    for i, (data, nv, enc, elem, bw, flags, dic) in enumerate(pages):
        descs[i].data = d_blob.data_ptr() + pos
        descs[i].data_len = len(data)
        descs[i].out = d_out.data_ptr() + opos
        descs[i].num_values = nv
        descs[i].encoding, descs[i].elem_size, descs[i].bit_width, descs[i].flags = enc, elem, bw, flags
        if dic is not None:
            descs[i].dict = (d_d32 if dic is d32 else d_d64).data_ptr()
            descs[i].dict_len = dic.size
        offs.append(opos)
        pos += len(data)
        opos += nv * elem
    results = (pq.PageResult * len(pages))()
    st = pq.lib.pqd_decode_pages(descs, results, len(pages), torch.cuda.current_stream().cuda_stream)
    assert st == 0, pq.lib.pqd_last_error()
    out = d_out.cpu().numpy()
    for i, ((data, nv, enc, elem, *_), exp) in enumerate(zip(pages, expect)):
        if exp is None:
            assert results[i].status == pq.BAD_ARG and results[i].values_written == 0
            continue
        assert results[i].status == 0 and results[i].values_written == nv, (i, enc, results[i].status)
        got = out[offs[i]:offs[i] + nv * elem].view(exp.dtype)
        assert (got == exp).all(), (i, enc)


def test_sharded_row_groups_decode_their_own_pages(orc):
    """The sharding of the multi-GPU workload driving real decodes: twelve row groups of mixed pages, assigned to
    1, 2 and 3 ranks; each rank's share is decoded with one pqd_decode_pages call per row group (all ranks
    take their turn on the one GPU of the test box) and every page is checked against the oracle."""
    import torch
    import parquet_java_amd as pq
    from parquet_java_amd.shard import assign_row_groups, decode_row_groups, gather_summaries
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(77)
    n = 5000
    d64 = rng.integers(-(1 << 62), 1 << 62, 3000).astype(np.int64)
    d_dict = gh.to_dev(d64.view(np.uint8))
    keep, groups, expect, sizes = [], [], [], []
    for g in range(12):
        pages = []
        for c in range(1 + g % 3):  # groups of different weight
            ids = rng.integers(0, d64.size, n).astype(np.int32)
            pages.append((orc.dict_encode_ids(ids, d64.size), pq.ENC_RLE_DICTIONARY, 8, 0, 0, d64[ids]))
            ts = np.cumsum(rng.integers(0, 1 << 10, n)).astype(np.int64)
            pages.append((orc.delta_encode(ts), pq.ENC_DELTA_BINARY_PACKED, 8, 0, 0, ts))
            lv = (rng.random(n) < 0.8).astype(np.int32)
            pages.append((orc.rle_encode(1, lv), pq.ENC_RLE, 4, 1, 0, lv))
            fl = rng.standard_normal(n).astype(np.float32)
            pages.append((fl.tobytes(), pq.ENC_PLAIN, 4, 0, 0, fl.view(np.int32)))
        blob = gh.to_dev(np.concatenate([np.frombuffer(bytes(p[0]), np.uint8) for p in pages]))
        out = torch.zeros(sum(n * p[2] for p in pages) + 64, dtype=torch.uint8, device=gh.dev())
        descs = (pq.PageDesc * len(pages))()
        pos = opos = 0
        where = []
        for i, (data, enc, elem, bw, flags, exp) in enumerate(pages):
            descs[i].data, descs[i].data_len, descs[i].out, descs[i].num_values = blob.data_ptr() + pos, len(bytes(data)), out.data_ptr() + opos, n
            descs[i].encoding, descs[i].elem_size, descs[i].bit_width, descs[i].flags = enc, elem, bw, flags
            if enc == pq.ENC_RLE_DICTIONARY:
                descs[i].dict, descs[i].dict_len = d_dict.data_ptr(), d64.size
            where.append((opos, n * elem, exp))
            pos += len(bytes(data))
            opos += n * elem
        keep.append((blob, out))
        groups.append(descs)
        expect.append(where)
        sizes.append(int(blob.numel()))
    stream = torch.cuda.current_stream().cuda_stream
    for world, policy in ((1, "round_robin"), (2, "round_robin"), (3, "greedy")):
        for _, out in keep:
            out.zero_()
        owned = assign_row_groups(sizes, world, policy)
        assert sorted(g for part in owned for g in part) == list(range(12))
        for rank in range(world):
            done = decode_row_groups(groups, rank, world, stream, sizes, policy)
            assert sorted(done) == owned[rank]
            summaries = gather_summaries([s for _, s in done.values()])  # no process group here: the local list
            assert all(not s["failed"] and s["values"] == s["pages"] * n for s in summaries[0])
        torch.cuda.synchronize()
        for (blob, out), where in zip(keep, expect):
            host = out.cpu().numpy()
            for opos, nbytes, exp in where:
                assert (host[opos:opos + nbytes].view(exp.dtype) == exp).all()
