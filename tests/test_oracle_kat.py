"""Pins the CPU oracle against the known-answer vectors and deterministic patterns held by
parquet-java's own tests (no GPU needed).

Each test names the reference test it transcribes; paths are relative to the parquet-java tree.
"""
import struct

import numpy as np
import pytest

from javarand import JavaRandom


def bits(s):
    return np.array([int(b, 2) for b in s.split()], dtype=np.uint8)


# ---------------------------------------------------------------------------
# independent model of the two bit orders: value i = bits [i*w, (i+1)*w) of the stream read
# as one little-endian integer (LE), or of the big-endian view (BE)
def model_unpack(w, data, n, msb_first):
    data = bytes(data)
    if msb_first:
        big = int.from_bytes(data, "big")
        total = len(data) * 8
        return [(big >> (total - (i + 1) * w)) & ((1 << w) - 1) for i in range(n)]
    big = int.from_bytes(data, "little")
    return [(big >> (i * w)) & ((1 << w) - 1) for i in range(n)]


@pytest.mark.parametrize("msb", [False, True])
def test_unpack8_matches_bit_order_model_int(orc, msb):
    rng = np.random.default_rng(1)
    for w in range(1, 33):
        for _ in range(20):
            data = rng.integers(0, 256, w, dtype=np.uint8)
            got = orc.unpack8(w, data, msb_first=msb).astype(np.int64) & 0xFFFFFFFF
            assert got.tolist() == model_unpack(w, data, 8, msb), (w, msb)


@pytest.mark.parametrize("msb", [False, True])
def test_unpack8_matches_bit_order_model_long(orc, msb):
    rng = np.random.default_rng(2)
    for w in range(1, 65):
        for _ in range(10):
            data = rng.integers(0, 256, w, dtype=np.uint8)
            got = [int(x) & 0xFFFFFFFFFFFFFFFF for x in orc.unpack8(w, data, msb_first=msb, long=True)]
            assert got == model_unpack(w, data, 8, msb), (w, msb)


def test_pack_unpack_round_trip_like_TestByteBitPacking(orc):
    # parquet-encoding/src/test/.../TestByteBitPacking.java:37-67,103-111 (BE int w 1-31, BE long w 1-63, Random(0))
    for w in range(1, 32):
        vals = np.array([(i * 7919 + 13) % (1 << w) for i in range(32)], np.int32)
        for msb in (False, True):
            packed = orc.pack(w, vals, msb_first=msb)
            assert packed.size == 4 * w
            out = np.concatenate([orc.unpack8(w, packed[g * w:(g + 1) * w], msb_first=msb) for g in range(4)])
            assert (out == vals).all()
    r = JavaRandom(0)
    for w in range(1, 64):
        vals = np.array([r.next_long() & ((1 << w) - 1) for _ in range(32)], np.uint64).astype(np.int64)
        for msb in (False, True):
            packed = orc.pack(w, vals, msb_first=msb, long=True)
            out = np.concatenate(
                [orc.unpack8(w, packed[g * w:(g + 1) * w], msb_first=msb, long=True) for g in range(4)])
            assert (out == vals).all()


def lemire_pack32(w, vals, msb_first):
    """Model of the int-based (Lemire) packer: 32 values -> w ints, serialised in the matching byte order
    (parquet-generator IntBasedBitPackingGenerator.java:99-160)."""
    words = []
    total = 0
    for i, v in enumerate(vals):
        total |= (int(v) & ((1 << w) - 1)) << (i * w) if not msb_first else 0
    if not msb_first:
        for k in range(w):
            words.append((total >> (32 * k)) & 0xFFFFFFFF)
        return b"".join(struct.pack("<I", x) for x in words)
    big = 0
    for v in vals:
        big = (big << w) | (int(v) & ((1 << w) - 1))
    return big.to_bytes(4 * w, "big")


def test_byte_packer_equals_lemire_int_packer(orc):
    # TestByteBitPacking.java:149-198: byte packer output == int packer output, both endiannesses, w 1-31
    for w in range(1, 32):
        vals = np.array([(i * 2654435761) % (1 << w) for i in range(32)], np.int64).astype(np.int32)
        for msb in (False, True):
            assert bytes(orc.pack(w, vals, msb_first=msb)) == lemire_pack32(w, vals, msb)


# ---------------------------------------------------------------------------
# parquet-encoding/src/test/.../TestBitPacking.java:35-155 and
# parquet-column/src/test/.../bitpacking/TestBitPackingColumn.java:37-156 (MSB-first bit strings)
BE_KATS = [
    (1, [0], "00000000"),
    (1, [1], "10000000"),
    (1, [0, 0], "00000000"),
    (1, [1, 1], "11000000"),
    (1, [1] * 9, "11111111 10000000"),
    (1, [0] * 9, "00000000 00000000"),
    (1, [0, 0, 0, 0, 0, 0, 0, 1], "00000001"),
    (1, [0] * 9 + [1], "00000000 01000000"),
    (1, [0, 1, 0, 0, 1, 1, 1, 0, 0, 1], "01001110 01000000"),
    (2, [0, 1, 2, 3, 3, 3, 2, 1, 1, 0, 0, 0, 1], "00011011 11111001 01000000 01000000"),
    (3, [0, 1, 2, 3, 4, 5, 6, 7, 1], "00000101 00111001 01110111 00100000"),
    (4, list(range(16)) + [1],
     "00000001 00100011 01000101 01100111 10001001 10101011 11001101 11101111 00010000"),
    (5, list(range(32)) + [1],
     "00000000 01000100 00110010 00010100 11000111 01000010 01010100 10110110 00110101 11001111 "
     "10000100 01100101 00111010 01010110 11010111 11000110 01110101 10111110 01110111 11011111 00001000"),
    (6, [0, 28, 34, 35, 63, 1], "00000001 11001000 10100011 11111100 00010000"),
    (7, [0, 28, 34, 35, 63, 1, 125, 1, 1], "00000000 01110001 00010010 00110111 11100000 01111110 10000001 00000010"),
]


@pytest.mark.parametrize("w,vals,expected", BE_KATS)
def test_bitpacked_levels_known_answers(orc, w, vals, expected):
    data = bits(expected)
    st, out, used = orc.bitpacked_levels_decode(w, data, len(vals))
    assert st == orc.OK
    assert out.tolist() == vals
    assert used == min((len(vals) * w + 7) // 8, data.size)
    # and the packer produces exactly those bytes for whole groups (ByteBitPackingValuesWriter pads with zeros)
    padded = list(vals) + [0] * (-len(vals) % 8)
    packed = bytes(orc.pack(w, np.array(padded, np.int32), msb_first=True))
    assert packed[:data.size] == bytes(data)


def test_bitpacked_levels_width_zero(orc):
    st, out, used = orc.bitpacked_levels_decode(0, b"", 10)  # TestBitPacking.testZero
    assert st == orc.OK and out.tolist() == [0] * 10 and used == 0


# ---------------------------------------------------------------------------
# parquet-column/src/test/.../rle/TestRunLengthBitPackingHybridEncoder.java:44-299
def test_rle_encoder_known_headers(orc):
    enc = bytes(orc.rle_encode(3, [4] * 100 + [5] * 100))  # testRLEOnly
    assert enc == bytes([200, 1, 4, 200, 1, 5])
    assert bytes(orc.rle_encode(3, [0] * 10)) == bytes([20, 0])  # testRepeatedZeros
    assert bytes(orc.rle_encode(0, [0] * 10)) == bytes([20])  # testBitWidthZero

    vals = [i % 3 for i in range(100)]  # testBitPackingOnly
    enc = orc.rle_encode(3, vals)
    assert enc[0] == 27 and enc.size == 1 + 13 * 3
    st, out, used = orc.read_batch(3, enc[1:], 104)
    assert st == orc.OK and out[:100].tolist() == vals and out[100:].tolist() == [0] * 4

    vals = [i % 3 for i in range(1000)]  # testBitPackingOverflow
    enc = orc.rle_encode(3, vals)
    assert enc[0] == 127 and enc[1 + 63 * 3] == 125 and enc.size == 2 + 125 * 3
    st, a, _ = orc.read_batch(3, enc[1:1 + 189], 504)
    st, b, _ = orc.read_batch(3, enc[2 + 189:], 496)
    assert np.concatenate([a, b]).tolist() == vals

    enc = bytes(orc.rle_encode(3, [0, 1, 0, 1, 0, 2, 2, 2] + [2] * 100))  # testTransitionFromBitPackingToRle
    assert enc[0] == 3 and enc[4:] == bytes([200, 1, 2])
    assert orc.read_batch(3, enc[1:4], 8)[1].tolist() == [0, 1, 0, 1, 0, 2, 2, 2]

    enc = orc.rle_encode(5, [i + 1 for i in range(9)])  # testPaddingZerosOnUnfinishedBitPackedRuns
    assert enc[0] == 5 and enc.size == 11
    assert orc.read_batch(5, enc[1:], 16)[1].tolist() == [1, 2, 3, 4, 5, 6, 7, 8, 9, 0, 0, 0, 0, 0, 0, 0]

    # testSwitchingModes (w = 9)
    vals = [17] * 25 + [7] * 7 + [8, 9, 10] + [6] * 25 + [5] * 8
    enc = bytes(orc.rle_encode(9, vals))
    assert enc[:3] == bytes([50, 17, 0])
    assert enc[3] == 5
    assert orc.read_batch(9, enc[4:22], 16)[1].tolist() == [7] * 7 + [8, 9, 10] + [6] * 6
    assert enc[22:] == bytes([38, 6, 0, 16, 5, 0])


def test_rle_decoder_group_boundary_kat(orc):
    # testGroupBoundary: {0x03, 1|2<<2|3<<4} at width 2 -> 1,2,3 although the run is truncated
    data = bytes([(1 << 1) | 1, (1 << 0) | (2 << 2) | (3 << 4)])
    st, out, used = orc.rle_decode(2, data, 3)
    assert st == orc.OK and out.tolist() == [1, 2, 3] and used == 2
    st, out, used = orc.rle_decode(2, data, 8)  # the rest of the group is zero-filled
    assert st == orc.OK and out.tolist() == [1, 2, 3, 0, 0, 0, 0, 0]
    st, out, used = orc.rle_decode(2, data, 9)  # the ninth read starts a new run at end of stream
    assert st == orc.PAST_END and out.size == 8


def integration_pattern(w):
    # RunLengthBitPackingHybridIntegrationTest.java:37-58
    m = 1 << w
    v = [i % m for i in range(100)] + [77 % m] * 100 + [88 % m] * 100
    for i in range(1000):
        v += [i % m] * 3
    v += [17 % m] * 1000
    return np.array(v, np.int64).astype(np.uint32).view(np.int32)


@pytest.mark.parametrize("w", range(0, 33))
def test_rle_integration_round_trip(orc, w):
    vals = integration_pattern(w)
    enc = orc.rle_encode(w, vals)
    st, out, used = orc.rle_decode(w, enc, vals.size)
    assert st == orc.OK and used == enc.size
    assert (out == vals).all()


def test_rle_value_not_masked_and_quirks(orc):
    # BytesUtils.readIntLittleEndianPaddedOnBitWidth reads whole bytes: an RLE value is not masked to w
    st, out, _ = orc.rle_decode(3, bytes([10 << 1, 0xFF]), 10)
    assert st == orc.OK and out.tolist() == [255] * 10
    # width > 32 is rejected by the decoder constructor
    assert orc.rle_decode(33, b"\x02\x00", 1)[0] == orc.BAD_ARG
    # zero-count RLE run: currentCount goes negative, the value repeats without another header read
    st, out, used = orc.rle_decode(8, bytes([0, 7, 4, 9]), 5)
    assert st == orc.OK and out.tolist() == [7] * 5 and used == 2
    # zero-group packed run: empty buffer is indexed -> failure, nothing produced
    st, out, _ = orc.rle_decode(8, bytes([1, 7]), 2)
    assert st == orc.DECODE and out.size == 0
    # truncated RLE value
    assert orc.rle_decode(16, bytes([4, 1]), 2)[0] == orc.EOF
    # multi-byte varint header: 200 values
    st, out, _ = orc.rle_decode(1, bytes([0x90, 0x03, 1]), 200)
    assert st == orc.OK and out.tolist() == [1] * 200
    # headers longer than five bytes (BytesUtils.readUnsignedVarInt, BytesUtils.java:202-211): the loop
    # reads until a byte without the continuation bit and Java shifts an int by (i & 31).  Worked by
    # hand: 0x94 then continuation bytes of zero payload then 0x00 is still 20 = RLE run of 10; five
    # 0x80 and then 0x05 puts the 5 at shift 35 & 31 = 3: header 40 = RLE run of 20.
    st, out, used = orc.rle_decode(8, bytes([0x94] + [0x80] * 5 + [0x00, 9]), 10)
    assert st == orc.OK and out.tolist() == [9] * 10 and used == 8
    st, out, used = orc.rle_decode(8, bytes([0x80] * 5 + [0x05, 9]), 40)
    assert st == orc.PAST_END and out.tolist() == [9] * 20 and used == 7
    assert orc.rle_decode(8, bytes([0x94] + [0x80] * 20), 4)[0] == orc.EOF


def test_rle_prefixed(orc):
    body = orc.rle_encode(2, [1, 2, 3, 0] * 10)
    data = struct.pack("<i", body.size) + bytes(body) + b"\xAA\xBB"
    st, out, used = orc.rle_decode(2, data, 40, prefixed=True)
    assert st == orc.OK and out.tolist() == [1, 2, 3, 0] * 10 and used == 4 + body.size
    assert orc.rle_decode(2, data[:8], 40, prefixed=True)[0] == orc.EOF


# ---------------------------------------------------------------------------
# ParquetReadRouter: readBatchUsing512Vector == readBatch == scalar unpack8 for every width
@pytest.mark.parametrize("w", range(0, 33))
def test_read_batch_all_widths(orc, w):
    rng = np.random.default_rng(100 + w)
    for n in (0, 8, 64, 72, 520, 2048):
        vals = rng.integers(0, 1 << w, n, dtype=np.uint64).astype(np.uint32).view(np.int32) if w else np.zeros(n, np.int32)
        packed = orc.pack(w, vals) if w else np.zeros(0, np.uint8)
        st, out, used = orc.read_batch(w, packed, n)
        assert st == orc.OK and used == n * w // 8
        assert (out == vals).all()
        if n and w:
            assert orc.read_batch(w, packed[:-1], n)[0] == orc.EOF
    # TestParquetReadRouter.java:41-57 input shape: byte i = (byte) i
    data = np.arange(0, 2048, dtype=np.uint64).astype(np.uint8)
    n = (data.size * 8 // w) // 8 * 8 if w else 64
    st, out, _ = orc.read_batch(w, data, n)
    assert st == orc.OK
    if w:
        assert (out.astype(np.int64) & 0xFFFFFFFF).tolist() == model_unpack(w, data, n, False)
    assert orc.read_batch(w, data, 12)[0] == orc.BAD_ARG


# ---------------------------------------------------------------------------
# dictionary pages: TestDictionary.java:347-379 (i % 50) and NaN bit patterns (:475-505)
def test_dictionary_round_trip(orc):
    ids = np.array([i % 50 for i in range(2000)], np.int32)
    page = orc.dict_encode_ids(ids, 50)
    assert page[0] == 6  # 32 - nlz(49)
    st, got = orc.dict_decode_ids(page, ids.size)
    assert st == orc.OK and (got == ids).all()
    d64 = (np.arange(50, dtype=np.int64) * 1000003) ^ 0x5555
    st, vals = orc.dict_decode(page, d64, ids.size)
    assert st == orc.OK and (vals == d64[ids]).all()
    nan_bits = np.array([0x7FC00000, 0x7FC00001, 0xFFC00000, 0x7F800001, 0x80000000, 0], np.uint32)
    page = orc.dict_encode_ids(np.arange(6, dtype=np.int32).repeat(3), 6)
    st, vals = orc.dict_decode(page, nan_bits.view(np.float32), 18)
    assert st == orc.OK and (vals.view(np.uint32) == nan_bits.repeat(3)).all()
    # id beyond the dictionary -> flagged, values before it delivered
    st, vals = orc.dict_decode(orc.dict_encode_ids(np.array([0, 1, 2, 9, 1], np.int32), 10), d64[:5], 5)
    assert st == orc.DICT_OOB and vals.size == 3
    # empty page: only an error once a value is asked for
    assert orc.dict_decode(b"", d64, 0)[0] == orc.OK
    assert orc.dict_decode(b"", d64, 1)[0] == orc.DECODE


# ---------------------------------------------------------------------------
# delta: DeltaBinaryPackingValuesWriterForLongTest.java:60-254, ...ForIntegerTest.java
def delta_cases_long():
    B = 128
    r = JavaRandom(0)
    yield "aligned", [r.next_long() for _ in range(5 * B)]
    r = JavaRandom(0)
    yield "block_not_full", [r.next_long() for _ in range(B - 3)]
    r = JavaRandom(0)
    yield "mini_not_full", [r.next_long() for _ in range(32 - 3)]
    r = JavaRandom(0)
    yield "negative_deltas", [10 - (i * 32 - r.next_int(6)) for i in range(B)]
    yield "same_deltas", [i * 32 for i in range(B)] + [0] * B
    yield "same_values", [3] * B + [0] * B
    yield "delta0_per_block", [int((i - 1) / B) for i in range(5 * B + 1)]  # Java int division truncates
    r = JavaRandom(0)
    yield "not_aligned", [r.next_int(20) - 10 for _ in range(5 * B + 3)]
    yield "min_max", [(-(1 << 63)) if i % 2 == 0 else (1 << 63) - 1 for i in range(10)]
    yield "offset", [i * 32 for i in range(2 * B + 3)]
    yield "one", [42]
    yield "none", []


@pytest.mark.parametrize("name,data", list(delta_cases_long()))
def test_delta_long_round_trips(orc, name, data):
    v = np.array(data, np.int64)
    page = orc.delta_encode(v, 128, 4)
    # size bound asserted by the reference test (:264-273)
    n = v.size
    if n:
        est = 15 + 10 + 8 * np.ceil((n - 1) / 32) * 32 + np.ceil((n - 1) / 128) * 4 + 10.0 * np.ceil((n - 1) / 128)
        assert est >= page.size
    junk = np.concatenate([page, np.full(7, 0xEE, np.uint8)])
    st, out, total, used = orc.delta_decode(junk, max(n, 100))
    assert st == orc.OK and total == n and used == page.size
    assert (out == v).all()


def test_delta_random_rounds_like_reference(orc):
    # randomDataTest (:240-254) with fewer rounds; the writer is reset between rounds
    r = JavaRandom(0)
    for _ in range(300):
        size = r.next_int(1000)
        v = np.array([r.next_long() for _ in range(size)], np.int64)
        page = orc.delta_encode(v)
        st, out, total, used = orc.delta_decode(page, 1000)
        assert st == orc.OK and total == size and used == page.size and (out == v).all()


def test_delta_int_writer(orc):
    r = JavaRandom(0)
    cases = [[r.next_int() for _ in range(5 * 128)], [r.next_int() for _ in range(125)],
             [(-(1 << 31)) if i % 2 == 0 else (1 << 31) - 1 for i in range(10)],
             [r.next_int(20) - 10 for _ in range(5 * 128 + 3)], [i * 32 for i in range(643)]]
    for data in cases:
        v = np.array(data, np.int32)
        page = orc.delta_encode(v)
        st, out, total, used = orc.delta_decode(page, v.size)
        assert st == orc.OK and total == v.size and used == page.size
        assert (out.astype(np.int32) == v).all()  # readInteger() = (int) readLong()


def test_delta_header_layout_and_errors(orc):
    page = bytes(orc.delta_encode(np.array([7, 8, 9], np.int64)))
    # block 128 -> varint 0x80 0x01, 4 miniblocks, 3 values, zigzag(7) = 14
    assert page[:5] == bytes([0x80, 0x01, 4, 3, 14])
    assert page[5] == 2 and page[6] == 0  # min delta 1 -> zigzag 2; first miniblock width 0
    assert orc.delta_decode(page[:3], 10)[0] == orc.EOF
    assert orc.delta_decode(bytes([0x80, 0x01, 4, 100, 14, 2, 8, 0, 0, 0, 1, 2]), 100)[0] == orc.EOF
    assert orc.delta_decode(bytes([100, 3, 3, 0]), 10)[0] == orc.BAD_ARG  # 33.3 values per miniblock
    assert orc.delta_decode(bytes([0x80, 0x01, 4, 40, 0, 0, 65, 0, 0, 0]), 100)[0] == orc.DECODE  # width 65
    assert orc.delta_decode(bytes([0x80, 0x01, 4, 40, 0]), 100)[0] == orc.DECODE  # no min delta


# ---------------------------------------------------------------------------
# byte stream split: ByteStreamSplitValuesReaderTest.java:60-102,175-230,248-297,331-359
def test_bss_known_answers(orc):
    st, out = orc.bss_decode(bytes([0x00, 0x00, 0x10, 0x40]), 4, 1)
    assert st == orc.OK and out.view(np.float32).tolist() == [2.25]
    data = bytes([0x40, 0x00, 0x80, 0x40, 0x05, 0x84, 0xC5, 0xBD, 0x32, 0xC2, 0x41, 0x42])
    st, out = orc.bss_decode(data, 4, 3)
    assert out.view(np.float32).tolist() == [-98.62548828125, 23.62744140625, 44.62939453125]
    st, out = orc.bss_decode(bytes([0xFE, 0xFF, 0xFF, 0x0D, 0xA8, 0x77, 0xD2, 0x40]), 8, 1)
    assert out.view(np.float64).tolist() == [18910.62585449218]
    data = bytes([0xE7, 0x72, 0xBE, 0x09, 0xA1, 0xC1, 0x0A, 0x0A, 0x17, 0xD7, 0x21, 0x26, 0x01, 0xC7, 0x53, 0x0A,
                  0x46, 0x05, 0x70, 0xF3, 0xE4, 0x40, 0xC0, 0x3F])
    st, out = orc.bss_decode(data, 8, 3)
    assert out.view(np.float64).tolist() == [256.625449218, -78956.4455667788, 0.62565]
    st, out = orc.bss_decode(bytes([0x12, 0x34, 0x56, 0x78, 0x9A, 0xBC, 0xDE, 0xF0]), 4, 2)
    assert out.view(np.uint32).tolist() == [0xDE9A5612, 0xF0BC7834]
    st, out = orc.bss_decode(bytes(range(0, 256, 17)), 8, 2)
    assert out.view(np.uint64).tolist() == [0xEECCAA8866442200, 0xFFDDBB9977553311]
    st, out = orc.bss_decode(bytes([0x12, 0x34, 0x56, 0x78, 0x9A, 0xBC]), 3, 2)  # FLBA(3)
    assert bytes(out) == bytes([0x12, 0x56, 0x9A, 0x34, 0x78, 0xBC])
    assert orc.bss_decode(bytes(7), 4, 10)[0] == orc.DECODE  # not a multiple of the element size
    assert orc.bss_decode(bytes(16), 4, 3)[0] == orc.DECODE  # more values than the page declares


def test_bss_random_like_reference(orc):
    r = JavaRandom(1337)
    f = np.array([np.float32(r.next_float()) * np.float32(1024.0) for _ in range(256)], np.float32)
    st, out = orc.bss_decode(orc.bss_encode(f.view(np.uint8), 4), 4, 256)
    assert st == orc.OK and (out.view(np.uint32) == f.view(np.uint32)).all()
    r = JavaRandom(6557)
    d = np.array([r.next_double() * 8192.0 for _ in range(256)], np.float64)
    st, out = orc.bss_decode(orc.bss_encode(d.view(np.uint8), 8), 8, 256)
    assert st == orc.OK and (out.view(np.uint64) == d.view(np.uint64)).all()
