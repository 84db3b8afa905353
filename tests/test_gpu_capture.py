"""pqd_decode_pages_dev captured into a HIP graph (after a warm-up call on the same stream, so that the
capture allocates nothing) and replayed: every replay must give what the eager call gives, also with eager
calls on the same stream in between.  A 70 000-page batch of tiny PLAIN pages (more pages than grid.y holds)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def prepare(pq, gh, torch, pages):
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
                          num_values=p.num_values, dict_len=dl, encoding=enc, elem_size=elem, bit_width=p.bit_width,
                          flags=p.flags))
    d_descs = gh.to_dev(pq.descs_to_numpy(pq.make_descs(descs)))
    d_res = torch.zeros(len(pages) * 16, dtype=torch.uint8, device=gh.dev())
    return dict(keep=(d_blob, dicts), descs=d_descs, res=d_res, out=d_out, n=len(pages), enc=enc, elem=elem)


def call(pq, b, stream):
    assert pq.lib.pqd_decode_pages_dev(b["descs"].data_ptr(), b["res"].data_ptr(), b["n"], b["enc"], b["elem"], stream.cuda_stream) == 0


def snapshot(torch, b):
    torch.cuda.synchronize()
    got = (b["res"].cpu().numpy().copy(), b["out"].cpu().numpy().copy())
    b["res"].zero_()
    b["out"].fill_(0xA5)
    torch.cuda.synchronize()
    return got


@pytest.mark.parametrize("kind", ["dict8", "dict4", "rle"])
def test_captured_call_replays_like_the_eager_one(orc, kind):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(17)

    def batch(n_pages):
        pages = []
        for k in range(n_pages):
            n = int(rng.integers(1, 30_000))
            if kind == "rle":
                lv = (rng.integers(0, 20, n) < 2).astype(np.int32) * rng.integers(0, 8, n).astype(np.int32)
                pages.append(gh.Page(orc.rle_encode(3, lv), n, pq.ENC_RLE, 4, bit_width=3))
                continue
            elem = 8 if kind == "dict8" else 4
            ids = rng.integers(0, 60_000, n).astype(np.int32)
            if k % 3 == 1:
                ids = np.sort(ids)
            pages.append(gh.Page(orc.dict_encode_ids(ids, 60_000), n, pq.ENC_RLE_DICTIONARY, elem, dictionary=dic[elem]))
        return pages

    dic = {8: rng.integers(-(1 << 62), 1 << 62, 60_000).astype(np.int64), 4: rng.integers(-(1 << 31), 1 << 31, 60_000).astype(np.int32)}
    a, other = prepare(pq, gh, torch, batch(200)), prepare(pq, gh, torch, batch(90))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        call(pq, a, s)  # warm-up on the stream the capture will use: its scratch exists from here on
        want_a = snapshot(torch, a)
        call(pq, other, s)
        want_other = snapshot(torch, other)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            call(pq, a, s)
        for rounds in range(3):
            g.replay()
            got = snapshot(torch, a)
            assert (got[0] == want_a[0]).all() and (got[1] == want_a[1]).all()
            g.replay()
            g.replay()  # twice in a row without an eager call in between
            got = snapshot(torch, a)
            assert (got[0] == want_a[0]).all() and (got[1] == want_a[1]).all()
            call(pq, other, s)  # an eager call between replays
            got = snapshot(torch, other)
            assert (got[0] == want_other[0]).all() and (got[1] == want_other[1]).all()
    pq.lib.pqd_stream_sync(s.cuda_stream)


def test_seventy_thousand_plain_pages_in_one_call(orc):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(3)
    n_pages, per = 70_000, 6
    vals = rng.integers(-(1 << 31), 1 << 31, n_pages * per).astype(np.int32)
    d_in = gh.to_dev(vals.view(np.uint8))
    d_out = torch.zeros(n_pages * per, dtype=torch.int32, device=gh.dev())
    d_descs = gh.to_dev(pq.descs_to_numpy(pq.make_descs(
        [dict(data=d_in.data_ptr() + i * per * 4, out=d_out.data_ptr() + i * per * 4, data_len=per * 4, num_values=per,
              encoding=pq.ENC_PLAIN, elem_size=4) for i in range(n_pages)])))
    d_res = torch.full((n_pages * 16,), 0x77, dtype=torch.uint8, device=gh.dev())
    assert pq.lib.pqd_decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), n_pages, pq.ENC_PLAIN, 4, None) == 0
    torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(np.int32).reshape(-1, 4)
    assert (res[:, 0] == 0).all() and (res[:, 1] == per).all()
    assert (d_out.cpu().numpy() == vals).all()
    # booleans the same way
    bits = rng.integers(0, 256, n_pages * 2, dtype=np.uint8)
    d_in = gh.to_dev(bits)
    d_out = torch.zeros(n_pages * 16, dtype=torch.uint8, device=gh.dev())
    d_descs = gh.to_dev(pq.descs_to_numpy(pq.make_descs(
        [dict(data=d_in.data_ptr() + i * 2, out=d_out.data_ptr() + i * 16, data_len=2, num_values=16,
              encoding=pq.ENC_PLAIN_BOOLEAN, elem_size=1) for i in range(n_pages)])))
    assert pq.lib.pqd_decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), n_pages, pq.ENC_PLAIN_BOOLEAN, 1, None) == 0
    torch.cuda.synchronize()
    assert (d_out.cpu().numpy() == np.unpackbits(bits, bitorder="little")).all()
