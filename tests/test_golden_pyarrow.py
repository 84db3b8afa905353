"""Column chunks written by pyarrow (tests/golden/pyarrow_chunks.json, made by
tests/golden/make_pyarrow_golden.py): an independent writer for the encodings the reference's tree holds
no byte vectors of.  DELTA_BINARY_PACKED (DeltaBinaryPackingValuesReader.java:60-172), BYTE_STREAM_SPLIT
(ByteStreamSplitValuesReader.java:53-218), RLE_DICTIONARY under V1 and V2 pages
(DictionaryValuesReader.java:49-127) and V1 / V2 definition levels (ColumnReaderBase.java:729-780) are
decoded by the oracle (CPU tests) and by the HIP library (GPU tests); both must give the values pyarrow
reads from its own file."""
import numpy as np
import pytest

import pagefile as pf

UNCOMPRESSED = [fx for fx in pf.load_fixtures() if fx["codec"] == "UNCOMPRESSED"]
VALUE_ENCODINGS = {"DELTA_BINARY_PACKED", "BYTE_STREAM_SPLIT", "RLE_DICTIONARY", "PLAIN_DICTIONARY"}


def wanted(fx):
    return fx["max_rep"] == 0 and bool(VALUE_ENCODINGS & set(fx["encodings"])) and fx["physical_type"] in pf.ELEM


CHUNKS = [fx for fx in UNCOMPRESSED if wanted(fx)]


class OracleBackend:
    def __init__(self, orc):
        self.orc = orc

    def levels(self, w, data, n, prefixed):
        st, out, used = self.orc.rle_decode(w, data, n, prefixed=prefixed)
        assert st == 0 and out.size == n
        return out, used

    def dictionary(self, data, dic, n):
        st, out = self.orc.dict_decode(data, dic, n)
        assert st == 0 and out.size == n
        return out

    def delta(self, data, n, elem):
        st, out, total, used = self.orc.delta_decode(data, n)
        assert st == 0 and total == n and used == len(data)
        return out.astype(np.int32) if elem == 4 else out

    def bss(self, data, k, n):
        st, out = self.orc.bss_decode(data, k, n)
        assert st == 0 and out.size == n * k
        return out


class GpuBackend:
    def __init__(self):
        import parquet_java_amd as pq
        import gpu_harness as gh
        pq.init(0)
        self.pq, self.gh = pq, gh

    def one(self, page):
        (st, written, consumed, out, guard), = self.gh.decode_pages([page])
        assert st == 0 and (guard == 0xA5).all()
        return written, consumed, out

    def levels(self, w, data, n, prefixed):
        written, consumed, out = self.one(self.gh.Page(data, n, self.pq.ENC_RLE, 4, bit_width=w,
                                                       flags=self.pq.F_LENGTH_PREFIX if prefixed else 0))
        assert written == n
        return out.view(np.int32), consumed

    def dictionary(self, data, dic, n):
        written, _, out = self.one(self.gh.Page(data, n, self.pq.ENC_RLE_DICTIONARY, dic.dtype.itemsize, dictionary=dic))
        assert written == n
        return out.view(dic.dtype)

    def delta(self, data, n, elem):
        written, consumed, out = self.one(self.gh.Page(data, n, self.pq.ENC_DELTA_BINARY_PACKED, elem))
        assert written == n and consumed == len(data)
        return out.view(np.int32 if elem == 4 else np.int64)

    def bss(self, data, k, n):
        written, consumed, out = self.one(self.gh.Page(data, n, self.pq.ENC_BYTE_STREAM_SPLIT, k))
        assert written == n and consumed == len(data)
        return out


def decode_chunk(fx, be):
    """Values (as unsigned bit patterns) and per-row presence of one flat column chunk."""
    elem = pf.ELEM[fx["physical_type"]]
    udt = np.dtype("<u%d" % elem)
    max_def = fx["max_def"]
    w = pf.level_width(max_def)
    dic = None
    values, present = [], []
    for page in pf.walk_chunk(pf.fixture_chunk(fx)):
        if page.type == pf.DICTIONARY_PAGE:
            assert page.encoding in ("PLAIN", "PLAIN_DICTIONARY")
            dic = np.frombuffer(page.body, udt, page.num_values).copy()
            continue
        n = page.num_values
        if page.type == pf.DATA_PAGE:
            _, d, body = pf.split_v1(page.body, 0, max_def, page.rep_encoding, page.def_encoding, n)
            prefixed = True
        else:
            _, d, body = pf.split_v2(page)
            prefixed = False
        n_present = n
        if max_def:
            lv, used = be.levels(w, np.frombuffer(d, np.uint8), n, prefixed)
            assert used == len(d)
            present.append(lv == max_def)
            n_present = int((lv == max_def).sum())
            if page.type == pf.DATA_PAGE_V2:
                assert n - n_present == page.num_nulls
        data = np.frombuffer(body, np.uint8)
        if page.encoding in ("RLE_DICTIONARY", "PLAIN_DICTIONARY"):
            values.append(be.dictionary(data, dic, n_present).view(udt))
        elif page.encoding == "DELTA_BINARY_PACKED":
            values.append(be.delta(data, n_present, elem).view(udt))
        elif page.encoding == "BYTE_STREAM_SPLIT":
            values.append(be.bss(data, elem, n_present).view(udt))
        else:
            values.append(np.frombuffer(body, udt, n_present))  # a PLAIN fallback page between dictionary pages
    return np.concatenate(values), (np.concatenate(present) if present else None)


def check_chunk(fx, be):
    got, present = decode_chunk(fx, be)
    expect = pf.fixture_values(fx).view(got.dtype)
    assert got.size == expect.size and (got == expect).all(), (fx["file"], fx["column"])
    want_present = pf.fixture_present(fx)
    if want_present is not None:
        assert (present == want_present).all(), (fx["file"], fx["column"])


def test_fixture_file_covers_the_encodings():
    encs = set()
    for fx in CHUNKS:
        encs |= set(fx["encodings"])
    assert {"DELTA_BINARY_PACKED", "BYTE_STREAM_SPLIT", "RLE_DICTIONARY"} <= encs
    assert {fx["data_page_version"] for fx in CHUNKS} == {"1.0", "2.0"}
    assert len(CHUNKS) >= 18


@pytest.mark.parametrize("fx", CHUNKS, ids=lambda fx: fx["file"] + "." + fx["column"])
def test_oracle_reads_pyarrow_written_chunks(orc, fx):
    check_chunk(fx, OracleBackend(orc))


@pytest.mark.gpu
@pytest.mark.parametrize("fx", CHUNKS, ids=lambda fx: fx["file"] + "." + fx["column"])
def test_gpu_reads_pyarrow_written_chunks(orc, fx, dict_path, rle_path):
    check_chunk(fx, GpuBackend())
