"""Pages written by parquet-java itself (cut out of the reference's test resources by
tests/golden/make_golden.py) decoded by the oracle (CPU) and by the HIP kernels (GPU)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def fixtures():
    with open(os.path.join(HERE, "golden", "java_pages.json")) as f:
        return json.load(f)["fixtures"]


def test_fixture_file_is_populated():
    fx = fixtures()
    kinds = {f["kind"] for f in fx}
    assert {"dictionary_ids", "def_levels", "rep_levels"} <= kinds and len(fx) >= 60
    assert any("1.9.0" in f["created_by"] for f in fx) and any("1.13.0" in f["created_by"] for f in fx)


def test_oracle_decodes_java_written_pages(orc):
    for f in fixtures():
        data = np.frombuffer(bytes.fromhex(f["data"]), np.uint8)
        if f["kind"] == "plain_values":
            st, vals = orc.plain_decode(data, f["elem_size"], f["num_values"])
            assert st == 0 and vals.tobytes().hex() == f["expected_hex"], f["column"]
            assert orc.plain_decode(data, f["elem_size"], f["num_values"] + 1)[0] == orc.DECODE  # "could not read ..."
            continue
        if f["kind"] == "plain_byte_array":  # tests/test_plain_binary.py
            continue
        exp = np.array(f["expected"], np.int32)
        if f["kind"] == "dictionary_ids":
            st, ids = orc.dict_decode_ids(data, exp.size)
            assert st == 0 and (ids == exp).all(), f["column"]
            # gather through a dictionary of the entry lengths (BINARY payloads themselves are out of scope)
            lens = np.array([len(s) for s in f["dictionary"]], np.int32)
            st, vals = orc.dict_decode(data, lens, exp.size)
            assert st == 0 and (vals == lens[exp]).all()
        elif f["encoding"] == "RLE":
            st, lv, used = orc.rle_decode(f["bit_width"], data, exp.size, prefixed=True)
            assert st == 0 and used == f["consumed"] and (lv == exp).all(), f["column"]
        else:
            st, lv, used = orc.bitpacked_levels_decode(f["bit_width"], data, exp.size)
            assert st == 0 and used == f["consumed"] and (lv == exp).all(), f["column"]


@pytest.mark.gpu
def test_gpu_decodes_java_written_pages(orc, dict_path, rle_path):
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    fx = fixtures()
    levels = [f for f in fx if f["kind"] in ("def_levels", "rep_levels") and f["encoding"] == "RLE"]
    pages = [gh.Page(bytes.fromhex(f["data"]), len(f["expected"]), pq.ENC_RLE, 4, bit_width=f["bit_width"],
                     flags=pq.F_LENGTH_PREFIX) for f in levels]
    for f, (st, written, consumed, out, guard) in zip(levels, gh.decode_pages(pages)):
        assert st == 0 and written == len(f["expected"]) and consumed == f["consumed"]
        assert out.view(np.int32).tolist() == f["expected"], f["column"]
    dicts = [f for f in fx if f["kind"] == "dictionary_ids"]
    pages = [gh.Page(bytes.fromhex(f["data"]), len(f["expected"]), pq.ENC_RLE, 4, flags=pq.F_WIDTH_PREFIX) for f in dicts]
    for f, (st, written, consumed, out, guard) in zip(dicts, gh.decode_pages(pages)):
        assert st == 0 and out.view(np.int32).tolist() == f["expected"], f["column"]
    for elem, dt in ((4, np.int32), (8, np.int64)):
        pages, exp = [], []
        for f in dicts:
            lens = np.array([len(s) * 1000003 + 7 for s in f["dictionary"]], dt)
            pages.append(gh.Page(bytes.fromhex(f["data"]), len(f["expected"]), pq.ENC_RLE_DICTIONARY, elem, dictionary=lens))
            exp.append(lens[np.array(f["expected"])])
        for e, (st, written, consumed, out, guard) in zip(exp, gh.decode_pages(pages)):
            assert st == 0 and written == e.size and (out.view(dt) == e).all()
    # PLAIN pages: INT32 / INT64 / DOUBLE values as parquet-java wrote them, one launch for all element sizes
    plain = [f for f in fx if f["kind"] == "plain_values"]
    assert len(plain) >= 20
    for elem in (4, 8):
        sel = [f for f in plain if f["elem_size"] == elem]
        pages = [gh.Page(bytes.fromhex(f["data"]), f["num_values"], pq.ENC_PLAIN, elem) for f in sel]
        pages += [gh.Page(bytes.fromhex(f["data"]), f["num_values"] + 3, pq.ENC_PLAIN, elem) for f in sel[:4]]  # reads past the end
        res = gh.decode_pages(pages)
        for f, (st, written, consumed, out, guard) in zip(sel, res):
            assert st == 0 and written == f["num_values"] and consumed == len(f["data"]) // 2
            assert out.tobytes().hex() == f["expected_hex"], f["column"]
            assert (guard == 0xA5).all()
        for f, (st, written, consumed, out, guard) in zip(sel[:4], res[len(sel):]):
            assert st == pq.DECODE and written == f["num_values"]
            assert out[:written * elem].tobytes().hex() == f["expected_hex"]
