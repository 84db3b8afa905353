"""pqd_decode_binary_pages_dev: dictionary data pages of a BINARY column -> offsets + bytes in one pass
(k_binary_fused.hip), against the oracle's restatement of the reference's read loop
(DictionaryValuesReader.readBytes over a PlainBinaryDictionary: DictionaryValuesReader.java:76-83,
PlainValuesDictionary.java:88-118): orc.dict_decode_ids gives the ids the loop gets before it fails,
orc.binary_materialize the bytes of those values."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_fused(pq, gh, torch, handle, pages, caps, fill=0xA5):
    """pages: [(page bytes, num_values)]; one call.  Returns per page
    (page status, page written, page consumed, bin status, bin written, bin bytes, offsets, data, guards)."""
    info = pq.DictInfo()
    pq.check(pq.lib.pqd_dict_get(handle, C.byref(info)))
    n = len(pages)
    blob = np.concatenate([np.full(1, 0xEE, np.uint8)] + [np.asarray(p[0], np.uint8) for p in pages] + [np.full(64, 0xEE, np.uint8)])
    d_blob = gh.to_dev(blob)
    at = np.cumsum([1] + [len(p[0]) for p in pages])
    o_pos = b_pos = 0
    lay = []
    for (data, nv), cap in zip(pages, caps):
        o_pos = (o_pos + 3) & ~3
        lay.append((o_pos + 4, b_pos + 3))
        o_pos += 4 + (nv + 1) * 4 + 12
        b_pos += 3 + cap + 13
    d_offs = torch.full((o_pos + 16,), fill, dtype=torch.uint8, device=gh.dev())
    d_bytes = torch.full((b_pos + 16,), fill, dtype=torch.uint8, device=gh.dev())
    pd = pq.make_descs([dict(data=d_blob.data_ptr() + int(at[i]), data_len=len(pages[i][0]), num_values=pages[i][1],
                             dict_len=info.n_entries, encoding=pq.ENC_RLE_DICTIONARY, elem_size=8) for i in range(n)])
    bd = (pq.BinaryPageDesc * n)()
    for i, ((data, nv), cap, (op, bp)) in enumerate(zip(pages, caps, lay)):
        bd[i].dict_bytes = info.bytes
        bd[i].offsets = d_offs.data_ptr() + op
        bd[i].bytes = d_bytes.data_ptr() + bp
        bd[i].dict_bytes_len = info.bytes_len
        bd[i].bytes_capacity = cap
    d_pd = gh.to_dev(pq.descs_to_numpy(pd))
    d_bd = gh.to_dev(np.frombuffer(bytes(bd), np.uint8).copy())
    d_pr = torch.full((n * 16,), 0x77, dtype=torch.uint8, device=gh.dev())
    d_br = torch.full((n * 16,), 0x77, dtype=torch.uint8, device=gh.dev())
    st = pq.lib.pqd_decode_binary_pages_dev(handle, d_pd.data_ptr(), d_bd.data_ptr(), d_pr.data_ptr(), d_br.data_ptr(), n,
                                            torch.cuda.current_stream().cuda_stream)
    assert st == 0, pq.lib.pqd_last_error()
    torch.cuda.synchronize()
    pr = d_pr.cpu().numpy().view(np.int32).reshape(-1, 4)
    br = d_br.cpu().numpy().view(np.int32).reshape(-1, 4)
    offs, data = d_offs.cpu().numpy(), d_bytes.cpu().numpy()
    out = []
    for i, ((pg, nv), cap, (op, bp)) in enumerate(zip(pages, caps, lay)):
        w = int(br[i, 1])
        guards = [offs[op - 4:op], offs[op + (w + 1) * 4:op + (nv + 1) * 4 + 12], data[bp - 3:bp], data[bp + cap:bp + cap + 13],
                  data[bp + int(br[i, 2]):bp + cap]]
        out.append((int(pr[i, 0]), int(pr[i, 1]), int(pr[i, 2]), int(br[i, 0]), w, int(br[i, 2]),
                    offs[op:op + (w + 1) * 4].view(np.int32).copy(), data[bp:bp + int(br[i, 2])].copy(), guards))
    return out


def expect(orc, page, nv, views, dict_page, cap):
    ist, ids = orc.dict_decode_ids(page, nv)
    usable = ids
    status = ist
    if ((ids < 0) | (ids >= views.size)).any():
        k = int(np.argmax((ids < 0) | (ids >= views.size)))
        usable, status = ids[:k], orc.DICT_OOB
    bst, off, data = orc.binary_materialize(views[usable], dict_page, cap)
    return status, usable.size, bst, off, data


def check(orc, pq, gh, torch, handle, views, dict_page, pages, caps):
    for (page, nv), cap, got in zip(pages, caps, run_fused(pq, gh, torch, handle, pages, caps)):
        pst, pwr, pused, bst, bwr, bbytes, offs, data, guards = got
        want_pst, want_n, want_bst, want_off, want_data = expect(orc, page, nv, views, dict_page, cap)
        assert bst == want_bst, (bst, want_bst, nv, cap)
        assert bwr == want_off.size - 1 and bbytes == want_data.size
        assert (offs == want_off).all()
        assert (data == want_data).all()
        if want_bst == 0:
            assert pst == want_pst and pwr == want_n, (pst, want_pst, pwr, want_n)
        else:  # the page stopped for lack of room or at a broken entry: the ids delivered so far
            assert pst == 0 and pwr == bwr
        for g in guards:
            assert (g == 0xA5).all()


def make_handle(pq, phys, page, n, type_length=0):
    h = C.c_void_p()
    page = np.ascontiguousarray(page, np.uint8)
    pq.check(pq.lib.pqd_dict_create(phys, type_length, page.ctypes.data_as(C.c_void_p), page.size, n, C.byref(h)))
    return h


@pytest.mark.parametrize("max_len,n_dict", [(12, 5000), (40, 3000), (12, 65536), (1, 300), (3000, 50)])
def test_fused_binary_pages_match_the_read_loop(orc, max_len, n_dict):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(max_len * 7 + n_dict)
    lo = 4 if n_dict == 65536 else 0
    values = [rng.integers(0, 256, int(rng.integers(lo, max_len + 1)), dtype=np.uint8).tobytes() for _ in range(n_dict)]
    dict_page = orc.binary_dict_page(values)
    st, views = orc.binary_dict_views(dict_page, n_dict)
    assert st == 0
    handle = make_handle(pq, pq.TYPE_BYTE_ARRAY, dict_page, n_dict)
    pages, caps = [], []
    for p in range(36 if max_len < 3000 else 10):
        n = int(rng.integers(1, 30000 if max_len < 3000 else 3000))
        if p == 0:
            n = 1
        ids = rng.integers(0, n_dict, n).astype(np.int32)
        if p % 4 == 1:
            ids = np.sort(ids)
        if p % 4 == 2:
            ids[rng.integers(0, n):] = ids[0]  # a long RLE run at the end
        if p % 9 == 3:
            ids[n // 2] = n_dict + 3  # an id beyond the dictionary part way
        page = orc.dict_encode_ids(ids, n_dict + 4)
        nv = n
        if p % 9 == 5:
            page = page[:max(2, page.size // 2)]  # the stream ends early
        if p % 9 == 7:
            nv = n + 100  # more values asked for than the stream holds
        total = int(sum(len(values[i]) for i in ids if i < n_dict))
        caps.append(total + 16 if p % 5 else max(0, total // 2))
        pages.append((page, nv))
    check(orc, pq, gh, torch, handle, views, dict_page, pages, caps)
    pq.lib.pqd_dict_destroy(handle)


def test_fused_fixed_length_and_broken_dictionaries(orc):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(4)
    # FIXED_LEN_BYTE_ARRAY of 5, 12, 13 and 16 bytes; INT96
    for phys, tl, fixed in ((pq.TYPE_FIXED_LEN_BYTE_ARRAY, 5, 5), (pq.TYPE_FIXED_LEN_BYTE_ARRAY, 12, 12), (pq.TYPE_FIXED_LEN_BYTE_ARRAY, 13, 13),
                            (pq.TYPE_FIXED_LEN_BYTE_ARRAY, 16, 16), (pq.TYPE_INT96, 0, 12)):
        n_dict = 777
        dict_page = rng.integers(0, 256, n_dict * fixed, dtype=np.uint8)
        st, views = orc.binary_dict_views(dict_page, n_dict, fixed)
        assert st == 0
        handle = make_handle(pq, phys, dict_page, n_dict, tl)
        pages, caps = [], []
        for p in range(6):
            n = int(rng.integers(1, 9000))
            ids = rng.integers(0, n_dict, n).astype(np.int32)
            pages.append((orc.dict_encode_ids(ids, n_dict), n))
            caps.append(n * fixed + (0 if p % 2 else 7))
        check(orc, pq, gh, torch, handle, views, dict_page, pages, caps)
        pq.lib.pqd_dict_destroy(handle)
    # a dictionary page whose last entries run past its end: they fail when a page refers to them
    values = [b"abc", b"defgh", b"", b"0123456789abcdefXYZ"]
    dict_page = orc.binary_dict_page(values)[:-6]
    st, views = orc.binary_dict_views(dict_page, 4)
    assert st == 0 and int(views[3]) == 0xFFFFFFFFFFFFFFFF
    handle = make_handle(pq, pq.TYPE_BYTE_ARRAY, dict_page, 4)
    pages = [(orc.dict_encode_ids(np.array(ids, np.int32), 4), len(ids)) for ids in ([0, 1, 2, 1, 0], [0, 1, 3, 1], [3], [2] * 50 + [3] * 20)]
    check(orc, pq, gh, torch, handle, views, dict_page, pages, [100] * 4)
    # pages that are empty, of no values, of another encoding
    got = run_fused(pq, gh, torch, handle, [(np.zeros(0, np.uint8), 5), (orc.dict_encode_ids(np.array([1, 1], np.int32), 4), 0),
                                            (np.array([40, 3, 0], np.uint8), 4)], [50, 50, 50])
    assert got[0][0] == orc.DECODE and got[0][4] == 0
    assert got[1][0] == 0 and got[1][3] == 0 and got[1][4] == 0 and got[1][6].tolist() == [0]
    assert got[2][0] == pq.BAD_ARG  # id width 40
    pq.lib.pqd_dict_destroy(handle)
