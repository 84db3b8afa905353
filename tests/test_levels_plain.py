"""PLAIN pages (fixed width, booleans) and the columnar use of decoded levels: validity bitmap and value
positions from definition levels, null scatter, list offsets from repetition levels.  The oracle
(oracle/pqd_oracle_pages.c restates PlainValuesReader.java:38-145, BooleanPlainValuesReader.java:38-89 and
the per-value loop of ColumnReaderBase.java:642-668) is checked against column chunks written by pyarrow
(values, null masks and list structure as pyarrow reads them back); the HIP kernels are checked against
the oracle and the same fixtures."""
import ctypes as C

import numpy as np
import pytest

import pagefile as pf

FIX = [fx for fx in pf.load_fixtures() if fx["codec"] == "UNCOMPRESSED"]
PLAIN = [fx for fx in FIX if fx["file"].startswith("plain_") and fx["physical_type"] in ("INT32", "INT64", "FLOAT", "DOUBLE", "BOOLEAN", "FIXED_LEN_BYTE_ARRAY")]
NESTED = [fx for fx in FIX if fx["max_rep"] == 1]


def chunk_pages(fx):
    """[(levels..., value section, page info)] of an uncompressed chunk, dictionary pages aside."""
    out = []
    for page in pf.walk_chunk(pf.fixture_chunk(fx)):
        if page.type == pf.DICTIONARY_PAGE:
            out.append(("dict", page))
            continue
        if page.type == pf.DATA_PAGE:
            r, d, body = pf.split_v1(page.body, fx["max_rep"], fx["max_def"], page.rep_encoding, page.def_encoding, page.num_values)
            out.append(("v1", page, r, d, body))
        else:
            r, d, body = pf.split_v2(page)
            out.append(("v2", page, r, d, body))
    return out


def elem_of(fx):
    return {"INT32": 4, "INT64": 8, "FLOAT": 4, "DOUBLE": 8, "BOOLEAN": 1, "FIXED_LEN_BYTE_ARRAY": fx["type_length"]}[fx["physical_type"]]


def expected_bytes(fx):
    v = pf.fixture_values(fx)
    if isinstance(v, list):
        return np.frombuffer(b"".join(v), np.uint8)
    return v.view(np.uint8)


def levels_of(orc, kind, sect, w, n):
    if w == 0:
        return np.zeros(n, np.int32)
    st, lv, used = orc.rle_decode(w, np.frombuffer(sect, np.uint8), n, prefixed=(kind == "v1"))
    assert st == 0 and used == len(sect)
    return lv


# ------------------------------------------------------------------------------- oracle vs pyarrow's files
@pytest.mark.parametrize("fx", PLAIN, ids=lambda fx: fx["file"] + "." + fx["column"])
def test_oracle_plain_pages_and_validity(orc, fx):
    elem = elem_of(fx)
    values, present = [], []
    for kind, page, r, d, body in chunk_pages(fx):
        n = page.num_values
        lv = levels_of(orc, kind, d, pf.level_width(fx["max_def"]), n) if fx["max_def"] else None
        validity, index, n_present = orc.def_validity(lv, fx["max_def"], n)
        present.append(np.unpackbits(validity, bitorder="little")[:n].astype(bool))
        assert (index[present[-1]] == np.arange(n_present)).all() and (index[~present[-1]] == -1).all()
        data = np.frombuffer(body, np.uint8)
        if fx["physical_type"] == "BOOLEAN":
            if page.encoding == "RLE":  # V2 booleans: RLE with a length prefix (RunLengthBitPackingHybridValuesReader)
                st, vals, used = orc.rle_decode(1, data, n_present, prefixed=True)
                assert st == 0
                vals = vals.astype(np.uint8)
            else:
                st, vals, used = orc.plain_boolean_decode(data, n_present)
                assert st == 0 and used == (n_present + 7) // 8
                assert orc.plain_boolean_decode(data, data.size * 8 + 1)[0] == orc.DECODE
        else:
            assert page.encoding == "PLAIN"
            st, vals = orc.plain_decode(data, elem, n_present)
            assert st == 0 and vals.size == n_present * elem
        values.append(vals)
    got = np.concatenate(values) if values else np.zeros(0, np.uint8)
    assert (got == expected_bytes(fx)).all()
    want_present = pf.fixture_present(fx)
    if want_present is not None:
        assert (np.concatenate(present) == want_present).all()
        # null scatter of the whole column: present slots hold the values in order, nulls are zero
        all_present = np.concatenate(present)
        st, spaced, written, bits = orc.scatter_nulls(got, np.packbits(all_present, bitorder="little"), all_present.size, elem)
        assert st == 0 and written == all_present.size and bits == all_present.sum()
        spaced = spaced.reshape(-1, elem)
        assert (spaced[all_present].reshape(-1) == got).all() and (spaced[~all_present] == 0).all()


@pytest.mark.parametrize("fx", NESTED, ids=lambda fx: fx["file"] + "." + fx["column"])
def test_oracle_list_offsets(orc, fx):
    import base64
    rows = np.frombuffer(base64.b64decode(fx["row_lengths"]), "<i4")  # -1: null list, else its length
    offs, valid, n_rows_total = [0], [], 0
    for entry in chunk_pages(fx):
        if entry[0] == "dict":
            continue
        kind, page, r, d, body = entry
        n = page.num_values
        rl = levels_of(orc, kind, r, 1, n)
        dl = levels_of(orc, kind, d, pf.level_width(fx["max_def"]), n)
        # optional list of optional int32: def 0 = null list, 1 = empty list, 2 = null element, 3 = element
        st, o, v, n_rows, n_slots = orc.rep_offsets(rl, dl, 2, 1, n)
        assert st == 0 and o.size == n_rows + 1
        offs += (o[1:] + offs[-1]).tolist()
        valid += v.tolist()
        if page.type == pf.DATA_PAGE_V2:
            assert n_rows == page.num_rows
    lengths = np.diff(np.array(offs))
    assert len(valid) == rows.size
    assert (np.array(valid, bool) == (rows >= 0)).all()
    assert (lengths == np.maximum(rows, 0)).all()


def test_oracle_level_assembly_known_answers(orc):
    # a record starts where the repetition level is 0; max definition level 3 = a present element
    rep = [0, 1, 1, 0, 0, 1, 0]
    dfl = [3, 2, 3, 0, 1, 3, 3]  # the 5th record is an empty list followed by... a level pair of its own
    st, o, v, rows, slots = orc.rep_offsets(rep, dfl, 2, 1, 10)
    assert st == 0 and rows == 4 and slots == 5 and o.tolist() == [0, 3, 3, 4, 5] and v.tolist() == [1, 0, 1, 1]
    st, o, v, rows, slots = orc.rep_offsets(rep, dfl, 2, 1, 2)  # room for two rows only
    assert st == orc.BAD_ARG and rows == 4 and o.tolist()[:2] == [0, 3]
    validity, index, n = orc.def_validity(dfl, 3)
    assert validity.tolist() == [0b1100101] and index.tolist() == [0, -1, 1, -1, -1, 2, 3] and n == 4
    validity, index, n = orc.def_validity(None, 0, 11)
    assert validity.tolist() == [0xFF, 0x07] and index.tolist() == list(range(11)) and n == 11
    st, out, written, bits = orc.scatter_nulls(np.array([7, 8], np.int32), np.array([0b1101], np.uint8), 4, 4)
    assert st == orc.DECODE and written == 3 and bits == 3 and out.view(np.int32).tolist() == [7, 0, 8, 0]


# ------------------------------------------------------------------------------- GPU
def dev_array(gh, a):
    return gh.to_dev(np.ascontiguousarray(a).view(np.uint8))


@pytest.mark.gpu
def test_gpu_plain_and_boolean_pages(orc):
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(61)
    for elem in (1, 2, 4, 5, 8, 12, 16):
        pages = []
        for n, extra, ask in ((0, 0, 0), (1, 0, 1), (7, 3, 7), (1000, 0, 1000), (20000, 5, 20000), (333, 0, 400), (100, elem - 1, 101)):
            data = rng.integers(0, 256, n * elem + extra, dtype=np.uint8)
            pages.append(gh.Page(data, ask, pq.ENC_PLAIN, elem))
        for p, (st, written, consumed, out, guard) in zip(pages, gh.decode_pages(pages)):
            ost, ovals = orc.plain_decode(p.data, elem, p.num_values)
            assert st == ost and written * elem == ovals.size and consumed == p.data.size
            assert (out[:ovals.size] == ovals).all() and (guard == 0xA5).all()
    for elem in (1, 4):
        pages = []
        for nbytes, ask in ((0, 0), (0, 3), (1, 8), (1, 5), (1, 9), (3, 24), (2500, 20000), (2500, 19997), (2499, 20000), (4000, 20000)):
            pages.append(gh.Page(rng.integers(0, 256, nbytes, dtype=np.uint8), ask, pq.ENC_PLAIN_BOOLEAN, elem))
        for p, (st, written, consumed, out, guard) in zip(pages, gh.decode_pages(pages)):
            ost, ovals, oused = orc.plain_boolean_decode(p.data, p.num_values)
            assert st == ost and written == ovals.size and consumed == oused
            got = out.view(np.uint8 if elem == 1 else np.int32)[:written]
            assert (got == ovals).all() and (guard == 0xA5).all()
    # the reader mirror: reads, skip, and the failures of the reference
    r = pq.PlainValuesReader.IntegerPlainValuesReader()
    vals = rng.integers(-1 << 31, 1 << 31, 10).astype(np.int32)
    r.initFromPage(10, pq.ByteBufferInputStream.wrap(vals.tobytes() + b"\x01\x02"))
    assert [r.readInteger() for _ in range(3)] == vals[:3].tolist()
    r.skip(5)
    assert [r.readInteger() for _ in range(2)] == vals[8:].tolist()
    with pytest.raises(pq.ParquetDecodingException):
        r.readInteger()
    b = pq.BooleanPlainValuesReader()
    b.initFromPage(10, pq.ByteBufferInputStream.wrap(bytes([0b10110010, 0b01, 0xFF])))
    assert [b.readBoolean() for _ in range(10)] == [False, True, False, False, True, True, False, True, True, False]
    assert b.getNextOffset() == 2
    with pytest.raises(pq.ParquetDecodingException):
        b.readBoolean()


def run_validity(pq, gh, torch, cases):
    """cases: (def_levels or None, n, max_def, want_index); one launch.  Returns per case (result fields, validity, index)."""
    keep, descs = [], (pq.ValidityDesc * len(cases))()
    outs = []
    for i, (lv, n, max_def, want_index) in enumerate(cases):
        d_lv = None if lv is None else dev_array(gh, np.ascontiguousarray(lv, np.int32))
        d_val = torch.full((((n + 7) // 8) + 3,), 0xA5, dtype=torch.uint8, device=gh.dev())
        d_idx = torch.full((n + 2,), -77, dtype=torch.int32, device=gh.dev()) if want_index else None
        keep.append((d_lv, d_val, d_idx))
        descs[i].def_levels = 0 if d_lv is None else d_lv.data_ptr()
        descs[i].validity = d_val.data_ptr() + 1  # an odd address
        descs[i].value_index = 0 if d_idx is None else d_idx.data_ptr()
        descs[i].n_levels = n
        descs[i].max_def = max_def
        outs.append((d_val, d_idx))
    d_descs = gh.to_dev(np.frombuffer(bytes(descs), np.uint8).copy())
    d_res = torch.zeros(len(cases) * 16, dtype=torch.uint8, device=gh.dev())
    assert pq.lib.pqd_def_levels_to_validity_dev(d_descs.data_ptr(), d_res.data_ptr(), len(cases), None) == 0
    torch.cuda.synchronize()
    raw = d_res.cpu().numpy().view(np.uint32).reshape(-1, 4)
    res = []
    for (lv, n, max_def, want_index), (d_val, d_idx), r in zip(cases, outs, raw):
        v = d_val.cpu().numpy()
        assert v[0] == 0xA5 and (v[1 + (n + 7) // 8:] == 0xA5).all()
        idx = None if d_idx is None else d_idx.cpu().numpy()
        if idx is not None:
            assert (idx[n:] == -77).all()
        res.append((int(r[0]), int(r[1]), int(r[3]), v[1:1 + (n + 7) // 8], None if idx is None else idx[:n]))
    return res


@pytest.mark.gpu
def test_gpu_validity_scatter_offsets_match_the_oracle(orc):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(62)
    cases = []
    for n in (0, 1, 7, 8, 9, 15, 16, 17, 4095, 4096, 4097, 20000, 100003):
        for max_def, p_null in ((1, 0.1), (3, 0.5), (2, 0.99), (1, 0.0)):
            lv = np.where(rng.random(n) < p_null, rng.integers(0, max(max_def, 1), n), max_def).astype(np.int32)
            cases.append((lv, n, max_def, n % 2 == 1 or n > 4096))
        cases.append((None, n, 0, True))
    res = run_validity(pq, gh, torch, cases)
    for (lv, n, max_def, want_index), (st, written, non_null, validity, index) in zip(cases, res):
        ov, oi, on = orc.def_validity(lv, max_def, n)
        assert st == 0 and written == n and non_null == on
        assert (validity == ov).all()
        if index is not None:
            assert (index == oi).all()
    # ---- null scatter, all element sizes, values short of the set bits in some cases
    for elem in (4, 8, 1, 3, 16):
        descs_in, keep = [], []
        for n in (0, 1, 16, 17, 5000, 20000, 70001):
            bits = rng.random(n) < rng.choice([0.0, 0.3, 0.9, 1.0])
            validity = np.packbits(bits, bitorder="little")
            n_set = int(bits.sum())
            for n_values in {n_set, max(n_set - 1, 0), n_set + 5}:
                values = rng.integers(1, 256, n_values * elem, dtype=np.uint8)
                descs_in.append((values, validity if n % 3 else (validity if n_set < n else None), n, n_values, bits))
        descs = (pq.ScatterDesc * len(descs_in))()
        for i, (values, validity, n, n_values, bits) in enumerate(descs_in):
            d_v = gh.to_dev(np.concatenate([values, np.zeros(16, np.uint8)]))
            d_b = None if validity is None else gh.to_dev(np.concatenate([validity, np.zeros(4, np.uint8)]))
            d_o = torch.full((n * elem + 8,), 0xA5, dtype=torch.uint8, device=gh.dev())
            keep.append((d_v, d_b, d_o))
            descs[i].values = d_v.data_ptr()
            descs[i].validity = 0 if d_b is None else d_b.data_ptr()
            descs[i].out = d_o.data_ptr()
            descs[i].n_slots = n
            descs[i].n_values = n_values
        d_descs = gh.to_dev(np.frombuffer(bytes(descs), np.uint8).copy())
        d_res = torch.zeros(len(descs_in) * 16, dtype=torch.uint8, device=gh.dev())
        assert pq.lib.pqd_scatter_nulls_dev(d_descs.data_ptr(), d_res.data_ptr(), len(descs_in), elem, None) == 0
        torch.cuda.synchronize()
        raw = d_res.cpu().numpy().view(np.uint32).reshape(-1, 4)
        for (values, validity, n, n_values, bits), (d_v, d_b, d_o), r in zip(descs_in, keep, raw):
            ost, oout, owritten, obits = orc.scatter_nulls(values, validity, n, elem)
            got = d_o.cpu().numpy()
            assert int(r[0]) == ost and int(r[1]) == owritten and int(r[3]) == obits, (elem, n, n_values)
            assert (got[:n * elem] == oout).all() and (got[n * elem:] == 0xA5).all()
    # ---- list offsets
    cases = []
    for n in (1, 2, 16, 17, 4096, 4097, 30000, 65537):
        for p_row in (0.05, 0.3, 1.0):
            rep = (rng.random(n) >= p_row).astype(np.int32)
            rep[0] = 0
            dfl = rng.integers(0, 4, n).astype(np.int32)
            n_rows = int((rep == 0).sum())
            cases.append((rep, dfl, n_rows))
            cases.append((rep, dfl, max(n_rows - 3, 0)))  # not enough room: PQD_BAD_ARG, the first rows are written
    descs = (pq.OffsetsDesc * len(cases))()
    keep = []
    for i, (rep, dfl, cap) in enumerate(cases):
        d_r, d_d = dev_array(gh, rep), dev_array(gh, dfl)
        d_o = torch.full((cap + 3,), -99, dtype=torch.int32, device=gh.dev())
        d_v = torch.full((cap + 4,), 0xA5, dtype=torch.uint8, device=gh.dev())
        keep.append((d_r, d_d, d_o, d_v))
        descs[i].rep_levels, descs[i].def_levels = d_r.data_ptr(), d_d.data_ptr()
        descs[i].offsets, descs[i].row_valid = d_o.data_ptr(), d_v.data_ptr() if i % 2 == 0 else 0
        descs[i].n_levels, descs[i].rows_capacity, descs[i].slot_def, descs[i].row_def = rep.size, cap, 2, 1
    d_descs = gh.to_dev(np.frombuffer(bytes(descs), np.uint8).copy())
    d_res = torch.zeros(len(cases) * 16, dtype=torch.uint8, device=gh.dev())
    assert pq.lib.pqd_rep_levels_to_offsets_dev(d_descs.data_ptr(), d_res.data_ptr(), len(cases), None) == 0
    torch.cuda.synchronize()
    raw = d_res.cpu().numpy().view(np.uint32).reshape(-1, 4)
    for i, ((rep, dfl, cap), (d_r, d_d, d_o, d_v), r) in enumerate(zip(cases, keep, raw)):
        ost, ooffs, ovalid, orows, oslots = orc.rep_offsets(rep, dfl, 2, 1, cap)
        assert int(r[0]) == ost and int(r[1]) == orows and int(r[3]) == oslots
        got = d_o.cpu().numpy()
        if ost == 0:
            assert (got[:orows + 1] == ooffs).all() and (got[orows + 1:] == -99).all()
        else:
            assert (got[:cap] == ooffs[:cap]).all() and (got[cap + 1:] == -99).all()
        if i % 2 == 0:
            v = d_v.cpu().numpy()
            assert (v[:min(orows, cap)] == ovalid).all() and (v[cap:] == 0xA5).all()


@pytest.mark.gpu
@pytest.mark.parametrize("fx", [f for f in PLAIN if f["physical_type"] != "BOOLEAN" or True], ids=lambda fx: fx["file"] + "." + fx["column"])
def test_gpu_reads_pyarrow_plain_chunks(orc, fx):
    """A whole column: levels (hybrid decode) -> validity -> PLAIN values -> null scatter, on the device;
    the spaced column must hold pyarrow's values at the non-null rows and zeros at the nulls."""
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    elem = elem_of(fx)
    out_elem = elem
    values, present = [], []
    for kind, page, r, d, body in chunk_pages(fx):
        n = page.num_values
        lv = None
        if fx["max_def"]:
            flags = pq.F_LENGTH_PREFIX if kind == "v1" else 0
            (st, written, consumed, out, _), = gh.decode_pages([gh.Page(d, n, pq.ENC_RLE, 4, bit_width=pf.level_width(fx["max_def"]), flags=flags)])
            assert st == 0 and written == n and consumed == len(d)
            lv = out.view(np.int32)
        (st, written, non_null, validity, _), = run_validity(pq, gh, torch, [(lv, n, fx["max_def"], False)])
        assert st == 0
        present.append(np.unpackbits(validity, bitorder="little")[:n].astype(bool))
        data = np.frombuffer(body, np.uint8)
        if fx["physical_type"] == "BOOLEAN" and page.encoding == "RLE":
            (st, written, _, out, _), = gh.decode_pages([gh.Page(data, non_null, pq.ENC_RLE, 4, bit_width=1, flags=pq.F_LENGTH_PREFIX)])
            vals = out.view(np.int32).astype(np.uint8)
        elif fx["physical_type"] == "BOOLEAN":
            (st, written, _, out, _), = gh.decode_pages([gh.Page(data, non_null, pq.ENC_PLAIN_BOOLEAN, 1)])
            vals = out
        else:
            (st, written, _, out, _), = gh.decode_pages([gh.Page(data, non_null, pq.ENC_PLAIN, elem)])
            vals = out
        assert st == 0 and written == non_null
        values.append(vals)
    got = np.concatenate(values) if values else np.zeros(0, np.uint8)
    assert (got == expected_bytes(fx)).all()
    all_present = np.concatenate(present)
    want_present = pf.fixture_present(fx)
    if want_present is not None:
        assert (all_present == want_present).all()
    # null scatter of the whole column in one call
    n = all_present.size
    d_v = gh.to_dev(np.concatenate([got, np.zeros(16, np.uint8)]))
    d_b = gh.to_dev(np.concatenate([np.packbits(all_present, bitorder="little"), np.zeros(4, np.uint8)]))
    d_o = torch.full((n * out_elem + 8,), 0xA5, dtype=torch.uint8, device=gh.dev())
    desc = (pq.ScatterDesc * 1)()
    desc[0].values, desc[0].validity, desc[0].out, desc[0].n_slots, desc[0].n_values = d_v.data_ptr(), d_b.data_ptr(), d_o.data_ptr(), n, got.size // elem
    d_descs = gh.to_dev(np.frombuffer(bytes(desc), np.uint8).copy())
    d_res = torch.zeros(16, dtype=torch.uint8, device=gh.dev())
    assert pq.lib.pqd_scatter_nulls_dev(d_descs.data_ptr(), d_res.data_ptr(), 1, out_elem, None) == 0
    torch.cuda.synchronize()
    spaced = d_o.cpu().numpy()[:n * elem].reshape(-1, elem)
    assert (spaced[all_present].reshape(-1) == got).all() and (spaced[~all_present] == 0).all()
    assert d_res.cpu().numpy().view(np.uint32)[0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("fx", NESTED, ids=lambda fx: fx["file"] + "." + fx["column"])
def test_gpu_list_offsets_of_pyarrow_chunks(orc, fx):
    import base64
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rows = np.frombuffer(base64.b64decode(fx["row_lengths"]), "<i4")
    offs, valid = [0], []
    for entry in chunk_pages(fx):
        if entry[0] == "dict":
            continue
        kind, page, r, d, body = entry
        n = page.num_values
        flags = pq.F_LENGTH_PREFIX if kind == "v1" else 0
        res = gh.decode_pages([gh.Page(r, n, pq.ENC_RLE, 4, bit_width=1, flags=flags),
                               gh.Page(d, n, pq.ENC_RLE, 4, bit_width=pf.level_width(fx["max_def"]), flags=flags)])
        assert all(x[0] == 0 and x[1] == n for x in res)
        d_r, d_d = dev_array(gh, res[0][3].view(np.int32)), dev_array(gh, res[1][3].view(np.int32))
        d_o = torch.zeros(n + 1, dtype=torch.int32, device=gh.dev())
        d_v = torch.zeros(n, dtype=torch.uint8, device=gh.dev())
        desc = (pq.OffsetsDesc * 1)()
        desc[0].rep_levels, desc[0].def_levels, desc[0].offsets, desc[0].row_valid = d_r.data_ptr(), d_d.data_ptr(), d_o.data_ptr(), d_v.data_ptr()
        desc[0].n_levels, desc[0].rows_capacity, desc[0].slot_def, desc[0].row_def = n, n, 2, 1
        d_descs = gh.to_dev(np.frombuffer(bytes(desc), np.uint8).copy())
        d_res = torch.zeros(16, dtype=torch.uint8, device=gh.dev())
        assert pq.lib.pqd_rep_levels_to_offsets_dev(d_descs.data_ptr(), d_res.data_ptr(), 1, None) == 0
        torch.cuda.synchronize()
        st, n_rows, _, n_slots = d_res.cpu().numpy().view(np.uint32)
        assert st == 0
        o = d_o.cpu().numpy()[:n_rows + 1]
        offs += (o[1:] + offs[-1]).tolist()
        valid += d_v.cpu().numpy()[:n_rows].tolist()
    assert (np.diff(np.array(offs)) == np.maximum(rows, 0)).all()
    assert (np.array(valid, bool) == (rows >= 0)).all()
