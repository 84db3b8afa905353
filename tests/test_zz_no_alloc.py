"""pqd_decode_pages_dev_noalloc (the entry point for callers that cannot allocate device memory)
against pqd_decode_pages_dev on the same pages, bit for bit, through the C ABI: two calls in a row,
the second with a smaller input, for dictionary pages of both element sizes and for plain hybrid
pages; then the limit and the pass-through encodings."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run(pq, gh, torch, pages, entry):
    """One launch over `pages` through `entry`; returns (status, written, consumed) arrays and the output bytes."""
    enc, elem = pages[0].encoding, pages[0].elem_size
    blob = np.concatenate([np.full(1, 0xEE, np.uint8)] + [p.data for p in pages] + [np.full(64, 0xEE, np.uint8)])
    d_blob = gh.to_dev(blob)
    offs = np.cumsum([1] + [p.data.size for p in pages])[:-1]
    out_offs = np.cumsum([0] + [((p.num_values * elem + 15) & ~15) + 16 for p in pages])
    d_out = torch.full((int(out_offs[-1]) + 16,), 0xA5, dtype=torch.uint8, device=gh.dev())
    dicts, descs = {}, []
    for p, o, oo in zip(pages, offs, out_offs):
        dd = dl = 0
        if p.dictionary is not None:
            if id(p.dictionary) not in dicts:
                dicts[id(p.dictionary)] = gh.to_dev(p.dictionary.view(np.uint8))
            dd, dl = dicts[id(p.dictionary)].data_ptr(), p.dictionary.size
        descs.append(dict(data=d_blob.data_ptr() + int(o), out=d_out.data_ptr() + int(oo), dict=dd, data_len=p.data.size,
                          num_values=p.num_values, dict_len=dl, encoding=p.encoding, elem_size=elem, bit_width=p.bit_width,
                          flags=p.flags))
    d_descs = gh.to_dev(pq.descs_to_numpy(pq.make_descs(descs)))
    d_res = torch.zeros(len(pages) * 16, dtype=torch.uint8, device=gh.dev())
    torch.cuda.synchronize()
    st = entry(d_descs.data_ptr(), d_res.data_ptr(), len(pages), enc, elem, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return st, pq.results_from_numpy(d_res.cpu().numpy()), d_out.cpu().numpy()


def dictionary_pages(pq, gh, orc, rng, n_pages, elem, n_dict):
    dt = np.int32 if elem == 4 else np.int64
    dic = rng.integers(-(1 << 31), 1 << 31, n_dict).astype(dt) * (3 if elem == 8 else 1)
    pages = []
    for k in range(n_pages):
        n = int(rng.integers(1, 25_000))
        ids = rng.integers(0, n_dict, n).astype(np.int32)
        if k % 4 == 1:
            ids = np.sort(ids)  # many short runs: several chunks per page
        if k % 4 == 2:
            ids[n // 2:] = ids[0]
        if k % 7 == 3:
            ids[n // 2] = n_dict + 5  # PQD_DICT_OOB part way
        pages.append(gh.Page(orc.dict_encode_ids(ids, n_dict + 6), n, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic))
    return pages


def level_pages(pq, gh, orc, rng, n_pages):
    pages = []
    for k in range(n_pages):
        n = int(rng.integers(1, 60_000))
        w = int(rng.integers(1, 9))
        lv = (rng.integers(0, 30, n) < 3).astype(np.int32) * rng.integers(0, 1 << w, n).astype(np.int32)
        if k % 3 == 0:
            lv = rng.integers(0, 1 << w, n).astype(np.int32)
        pages.append(gh.Page(orc.rle_encode(w, lv), n + (5 if k % 5 == 4 else 0), pq.ENC_RLE, 4, bit_width=w))
    return pages


def same(a, b):
    (sa, ra, oa), (sb, rb, ob) = a, b
    assert sa == sb == 0
    for x, y in zip(ra, rb):
        assert (x == y).all()
    assert (oa == ob).all()  # outputs, guard bytes between them included


def test_no_alloc_entry_matches_the_allocating_one(orc):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(2026)
    noalloc, alloc = pq.lib.pqd_decode_pages_dev_noalloc, pq.lib.pqd_decode_pages_dev
    for make in (lambda n: dictionary_pages(pq, gh, orc, rng, n, 8, 60_000),
                 lambda n: dictionary_pages(pq, gh, orc, rng, n, 4, 3_000),
                 lambda n: level_pages(pq, gh, orc, rng, n)):
        first, second = make(300), make(37)  # the second call is smaller: the scratch still holds the first one's chunks
        got1 = run(pq, gh, torch, first, noalloc)
        got2 = run(pq, gh, torch, second, noalloc)
        same(got1, run(pq, gh, torch, first, alloc))
        same(got2, run(pq, gh, torch, second, alloc))
        # and against the oracle, for the second call
        _, (status, written, _), out = got2
        pos = 0
        for p, st, wr in zip(second, status, written):
            if p.encoding == pq.ENC_RLE_DICTIONARY:
                ost, ovals = orc.dict_decode(p.data, p.dictionary, p.num_values)
            else:
                ost, ovals, _ = orc.rle_decode(p.bit_width, p.data, p.num_values)
            assert st == ost and wr == ovals.size
            assert (out[pos:pos + ovals.nbytes] == ovals.view(np.uint8)).all()
            pos += ((p.num_values * p.elem_size + 15) & ~15) + 16


def test_no_alloc_entry_limits_and_pass_through(orc, monkeypatch):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(7)
    noalloc, alloc = pq.lib.pqd_decode_pages_dev_noalloc, pq.lib.pqd_decode_pages_dev
    # the largest batch it takes, and one page more
    tiny = [gh.Page(orc.rle_encode(3, rng.integers(0, 8, 40).astype(np.int32)), 40, pq.ENC_RLE, 4, bit_width=3) for _ in range(2048)]
    same(run(pq, gh, torch, tiny, noalloc), run(pq, gh, torch, tiny, alloc))
    st, _, _ = run(pq, gh, torch, tiny + tiny[:1], noalloc)
    assert st == pq.BAD_ARG and b"2048" in pq.lib.pqd_last_error()
    # long pages that need more chunks than the pool holds (its size lowered for the test): what does
    # not fit goes to the generic kernel
    long_pages = [gh.Page(orc.rle_encode(7, (np.arange(1_000_000) % 100 + rng.integers(0, 3)).astype(np.int32)), 1_000_000,
                          pq.ENC_RLE, 4, bit_width=7) for _ in range(3)]
    want = run(pq, gh, torch, long_pages, alloc)
    monkeypatch.setenv("PQD_POOL_CHUNKS", "40")
    same(run(pq, gh, torch, long_pages, noalloc), want)
    monkeypatch.delenv("PQD_POOL_CHUNKS")
    same(run(pq, gh, torch, long_pages, noalloc), want)
    # encodings without scratch are passed through
    ts = np.cumsum(rng.integers(0, 9000, 5000)).astype(np.int64)
    delta = [gh.Page(orc.delta_encode(ts), ts.size, pq.ENC_DELTA_BINARY_PACKED, 8)]
    a = run(pq, gh, torch, delta, noalloc)
    same(a, run(pq, gh, torch, delta, alloc))
    assert (a[2][:ts.nbytes].view(np.int64) == ts).all()


def test_no_alloc_calls_on_two_streams_queue_up(orc):
    """Calls on different streams of one GPU share the GPU's scratch array: the library orders them on the
    device (an event behind every call), so both produce what they produce alone."""
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(99)
    noalloc, alloc = pq.lib.pqd_decode_pages_dev_noalloc, pq.lib.pqd_decode_pages_dev
    batches = [dictionary_pages(pq, gh, orc, rng, 400, 8, 60_000), dictionary_pages(pq, gh, orc, rng, 350, 4, 40_000),
               level_pages(pq, gh, orc, rng, 300), dictionary_pages(pq, gh, orc, rng, 64, 8, 60_000)]
    want = [run(pq, gh, torch, b, alloc) for b in batches]
    # the same batches, enqueued back to back on alternating streams with no host wait in between
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    prepared = []
    for b in batches:
        enc, elem = b[0].encoding, b[0].elem_size
        blob = np.concatenate([np.full(1, 0xEE, np.uint8)] + [p.data for p in b] + [np.full(64, 0xEE, np.uint8)])
        d_blob = gh.to_dev(blob)
        offs = np.cumsum([1] + [p.data.size for p in b])[:-1]
        out_offs = np.cumsum([0] + [((p.num_values * elem + 15) & ~15) + 16 for p in b])
        d_out = torch.full((int(out_offs[-1]) + 16,), 0xA5, dtype=torch.uint8, device=gh.dev())
        dicts, descs = {}, []
        for p, o, oo in zip(b, offs, out_offs):
            dd = dl = 0
            if p.dictionary is not None:
                if id(p.dictionary) not in dicts:
                    dicts[id(p.dictionary)] = gh.to_dev(p.dictionary.view(np.uint8))
                dd, dl = dicts[id(p.dictionary)].data_ptr(), p.dictionary.size
            descs.append(dict(data=d_blob.data_ptr() + int(o), out=d_out.data_ptr() + int(oo), dict=dd, data_len=p.data.size,
                              num_values=p.num_values, dict_len=dl, encoding=enc, elem_size=elem, bit_width=p.bit_width,
                              flags=p.flags))
        d_descs = gh.to_dev(pq.descs_to_numpy(pq.make_descs(descs)))
        d_res = torch.zeros(len(b) * 16, dtype=torch.uint8, device=gh.dev())
        prepared.append((d_blob, dicts, d_descs, d_res, d_out, len(b), enc, elem))
    torch.cuda.synchronize()
    for rounds in range(3):
        for k, (d_blob, dicts, d_descs, d_res, d_out, n, enc, elem) in enumerate(prepared):
            assert noalloc(d_descs.data_ptr(), d_res.data_ptr(), n, enc, elem, streams[k % 2].cuda_stream) == 0
    torch.cuda.synchronize()
    for (d_blob, dicts, d_descs, d_res, d_out, n, enc, elem), w in zip(prepared, want):
        same((0, pq.results_from_numpy(d_res.cpu().numpy()), d_out.cpu().numpy()), w)
