"""Page decompression (SNAPPY, LZ4_RAW, ZSTD, GZIP).  The algorithms live in third-party libraries the reference
calls (snappy-java 1.1.10.8, aircompressor 2.0.3, zstd-jni 1.5.7-13, Hadoop's GzipCodec over zlib); the oracle
restates the published formats (oracle/pqd_oracle_pages.c, pqd_oracle_zstd.c, pqd_oracle_gzip.c) and is pinned by
byte strings and parquet pages compressed with the C libraries of the same formats through pyarrow and Python's
zlib module (tests/golden/codec_vectors.json, gzip_vectors.json, pyarrow_chunks.json, pyarrow_gzip_chunks.json)
and by hand-made streams that use every element type; the HIP kernels are checked against the oracle on all of
these, on streams compressed when the test runs (large inputs, several compression levels) and on damaged
streams."""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

import codec_inputs
import pagefile as pf

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "codec_vectors.json")) as f:
    VECTORS = json.load(f)["vectors"]
with open(os.path.join(HERE, "golden", "gzip_vectors.json")) as f:
    VECTORS += json.load(f)["vectors"]
GZIP_FORMS = ("_l1", "_l6", "_l9", "_fixed", "_stored", "_huffman_only", "_three_members", "_header_fields")
INPUTS = dict(codec_inputs.inputs())
CODEC_NUM = {"SNAPPY": 1, "LZ4_RAW": 7, "ZSTD": 6, "UNCOMPRESSED": 0, "GZIP": 2}
PAGE_CODEC = {"SNAPPY": "SNAPPY", "LZ4": "LZ4_RAW", "ZSTD": "ZSTD", "GZIP": "GZIP"}  # (pyarrow's metadata calls LZ4_RAW "LZ4")
PAGES = [fx for fx in pf.load_fixtures() + pf.load_fixtures("pyarrow_gzip_chunks.json") if fx["codec"] in PAGE_CODEC]
CODECS = ["SNAPPY", "LZ4_RAW", "ZSTD", "GZIP"]


def base_name(name):
    """The input a vector's name refers to (ZSTD vectors are named input_l<level>, GZIP vectors input_<form>)."""
    for form in GZIP_FORMS:
        if name.endswith(form):
            return name[:-len(form)]
    return name


def original(v):
    """The input a vector was made from (ZSTD vectors are named input_l<level>)."""
    name = v["name"]
    return INPUTS[base_name(name) if v["codec"] in ("ZSTD", "GZIP") else name]


class _Originals:
    def __getitem__(self, name):
        return INPUTS[name] if name in INPUTS else INPUTS[base_name(name)]


ORIGINALS = _Originals()


def compressed_parts(fx):
    """[(compressed bytes, uncompressed size)] of the compressed parts of a chunk's pages."""
    out = []
    for page in pf.walk_chunk(pf.fixture_chunk(fx)):
        body, size = page.body, page.uncompressed_size
        if page.type == pf.DATA_PAGE_V2:
            if not page.is_compressed:
                continue
            levels = page.rep_len + page.def_len
            body, size = body[levels:], size - levels
        out.append((body, size))
    return out


def varint(n):
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def handmade_snappy():
    """(stream, original): every element type of the format description."""
    rng = np.random.default_rng(5)
    lit = lambda b: (bytes([(len(b) - 1) << 2]) if len(b) <= 60 else  # noqa: E731
                     bytes([60 << 2, len(b) - 1]) if len(b) <= 256 else
                     bytes([61 << 2]) + (len(b) - 1).to_bytes(2, "little") if len(b) <= 65536 else
                     bytes([62 << 2]) + (len(b) - 1).to_bytes(3, "little")) + b
    parts, out = [], bytearray()

    def add_lit(b):
        parts.append(lit(b))
        out.extend(b)

    def add_copy(offset, length, kind):
        if kind == 1:
            assert 4 <= length <= 11 and offset < 2048
            parts.append(bytes([1 | ((length - 4) << 2) | ((offset >> 8) << 5), offset & 0xFF]))
        elif kind == 2:
            parts.append(bytes([2 | ((length - 1) << 2)]) + offset.to_bytes(2, "little"))
        else:
            parts.append(bytes([3 | ((length - 1) << 2)]) + offset.to_bytes(4, "little"))
        for _ in range(length):
            out.append(out[-offset])

    add_lit(b"abcdefgh")
    add_copy(8, 8, 1)
    add_copy(1, 11, 1)      # a run
    add_copy(3, 64, 2)      # overlapping, period 3
    add_lit(bytes(rng.integers(0, 256, 61, dtype=np.uint8)))    # one length byte
    add_lit(bytes(rng.integers(0, 256, 300, dtype=np.uint8)))   # two length bytes
    add_lit(bytes(rng.integers(0, 256, 70000, dtype=np.uint8)))  # three length bytes
    add_copy(70000, 64, 3)  # four-byte offset, beyond 64 KiB
    add_copy(40000, 33, 2)  # beyond the 32 KiB the kernel keeps in LDS
    add_copy(65, 64, 2)
    add_copy(64, 64, 2)
    add_copy(63, 64, 2)
    add_lit(b"z")
    return varint(len(out)) + b"".join(parts), bytes(out)


def handmade_snappy_far_copies():
    """(stream, original): copies from far back right in front of long literals, and copies from about a ring
    size back in front of short ones.  A decoder that keeps the last 8 / 16 KiB of output in a ring and writes
    an element's literal bytes into it before it has run the copies in front of them reads the wrong bytes here."""
    rng = np.random.default_rng(55)
    parts, out = [], bytearray()

    def add_lit(b):
        n = len(b) - 1
        parts.append((bytes([n << 2]) if n < 60 else bytes([60 << 2, n]) if n < 256 else
                      bytes([61 << 2]) + n.to_bytes(2, "little") if n < 65536 else bytes([62 << 2]) + n.to_bytes(3, "little")) + b)
        out.extend(b)

    def add_copy(offset, length):
        parts.append(bytes([2 | ((length - 1) << 2)]) + offset.to_bytes(2, "little"))
        for _ in range(length):
            out.append(out[-offset])

    text = (b"the quick brown fox jumps over the lazy dog; " * 400)[:17000]
    add_lit(text)
    for offset, tail in ((12000, 6000), (16000, 3000), (8000, 9000), (16384, 700), (8192, 64), (7000, 1500),
                         (16380, 40), (8190, 30), (16300, 5), (8100, 5)):
        add_copy(offset, 64)
        add_copy(offset, 37)
        add_lit(bytes(rng.integers(0, 256, tail, dtype=np.uint8)))
        add_copy(offset // 2, 50)
        for _ in range(6):  # short elements, so that one 64-byte window of input holds several of them
            add_copy(offset - 11, 9)
            add_lit(bytes(rng.integers(0, 256, 3, dtype=np.uint8)))
        add_lit(text[:2000])
    return varint(len(out)) + b"".join(parts), bytes(out)


def handmade_lz4():
    rng = np.random.default_rng(6)
    parts, out = [], bytearray()

    def ext(n):
        b = bytearray()
        while n >= 255:
            b.append(255)
            n -= 255
        b.append(n)
        return bytes(b)

    def seq(literals, offset=None, mlen=None):
        ll = len(literals)
        ml = 0 if offset is None else mlen - 4
        token = (min(ll, 15) << 4) | min(ml, 15)
        s = bytes([token]) + (ext(ll - 15) if ll >= 15 else b"") + literals
        out.extend(literals)
        if offset is not None:
            s += offset.to_bytes(2, "little") + (ext(ml - 15) if ml >= 15 else b"")
            for _ in range(mlen):
                out.append(out[-offset])
        parts.append(s)

    seq(b"abcd", 4, 4)
    seq(b"", 1, 300)           # a run with extension bytes
    seq(bytes(rng.integers(0, 256, 15, dtype=np.uint8)), 3, 19)      # lengths exactly at the nibble limit
    seq(bytes(rng.integers(0, 256, 15 + 255, dtype=np.uint8)), 7, 19 + 255)  # an extension byte of 255 then 0
    seq(bytes(rng.integers(0, 256, 5000, dtype=np.uint8)), 5000, 70000)  # longer than the kernel's ring
    seq(bytes(rng.integers(0, 256, 40000, dtype=np.uint8)), 65535, 10)
    seq(b"", 33000, 100)
    seq(b"tail!")
    return b"".join(parts), bytes(out)


class BitWriter:
    """Deflate's bit order: fields least significant bit first, Huffman codes most significant bit first."""

    def __init__(self):
        self.bits = []

    def field(self, value, n):
        self.bits += [(value >> k) & 1 for k in range(n)]

    def code(self, value, n):
        self.bits += [(value >> k) & 1 for k in range(n - 1, -1, -1)]

    def align(self):
        self.bits += [0] * (-len(self.bits) % 8)

    def raw(self, data):
        self.align()
        for b in data:
            self.field(b, 8)

    def bytes(self):
        self.align()
        return bytes(sum(bit << k for k, bit in enumerate(self.bits[i:i + 8])) for i in range(0, len(self.bits), 8))


def fixed_literal(w, sym):
    """A literal / length symbol in the fixed code (RFC 1951 section 3.2.6)."""
    if sym < 144:
        w.code(0x30 + sym, 8)
    elif sym < 256:
        w.code(0x190 + sym - 144, 9)
    elif sym < 280:
        w.code(sym - 256, 7)
    else:
        w.code(0xC0 + sym - 280, 8)


def gzip_member(deflate, raw, flg=0, extra=b""):
    import zlib
    return bytes([0x1F, 0x8B, 8, flg, 0, 0, 0, 0, 0, 255]) + extra + deflate + zlib.crc32(raw).to_bytes(4, "little") + \
        (len(raw) & 0xFFFFFFFF).to_bytes(4, "little")


def handmade_gzip():
    """(stream, original): two members written bit by bit: an empty stored block, a fixed-code block with a match that
    repeats one byte 258 times and one that overlaps itself, a stored block; then a member of one empty fixed block
    and one whose dynamic code uses all three repeat codes and has a single distance code."""
    w = BitWriter()
    w.field(0, 1), w.field(0, 2), w.raw((0).to_bytes(2, "little") + (0xFFFF).to_bytes(2, "little"))  # stored, empty
    w.field(0, 1), w.field(1, 2)
    out = bytearray()
    for ch in b"abc":
        fixed_literal(w, ch)
        out.append(ch)
    fixed_literal(w, 257), w.code(2, 5)  # length 3, distance 3
    out += out[-3:]
    fixed_literal(w, 285), w.code(0, 5)  # length 258, distance 1
    out += bytes([out[-1]]) * 258
    fixed_literal(w, 265), w.field(1, 1), w.code(4, 5), w.field(1, 1)  # length 12, distance 6
    for _ in range(12):
        out.append(out[-6])
    fixed_literal(w, 256)
    w.field(1, 1), w.field(0, 2), w.raw((4).to_bytes(2, "little") + (0xFFFB).to_bytes(2, "little") + b"tail")
    out += b"tail"
    first = gzip_member(w.bytes(), bytes(out))
    # second member: an empty fixed block, then a dynamic block.  Literal / length lengths: 'x' (120) and 'y' (121) 2 bits,
    # end of block (256) 2 bits, length symbol 257 2 bits; one distance code of 1 bit (distance 1).
    w = BitWriter()
    w.field(0, 1), w.field(1, 2), fixed_literal(w, 256)
    w.field(1, 1), w.field(2, 2)
    w.field(258 - 257, 5), w.field(0, 5), w.field(19 - 4, 4)  # HLIT 258, HDIST 1, HCLEN 19
    # code-length code: symbols 0 (2 bits), 1 (3), 2 (3), 16 (3), 17 (3), 18 (2) -> complete
    cl = {0: 2, 1: 3, 2: 3, 16: 3, 17: 3, 18: 2}
    for sym in (16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15):
        w.field(cl.get(sym, 0), 3)
    codes, code = {}, 0
    for n in (2, 3):
        for sym in sorted(cl):
            if cl[sym] == n:
                codes[sym] = (code, n)
                code += 1
        code <<= 1
    def put(sym):  # noqa: E306
        w.code(*codes[sym])
    put(18), w.field(120 - 11, 7)           # 120 zeros
    put(2), put(16), w.field(0, 2)          # 'x' = 2 bits; repeat it three times: 'y', 122, 123 (too many: fixed below)
    # (the repeat above gave 121..123 two bits each; that is one code too many for a complete set, so take it back:
    #  the stream is rebuilt with a repeat of exactly one further symbol, which code 16 cannot do: spell it out)
    w2 = BitWriter()
    w2.bits = w.bits[:-(3 + 2 + 3)]          # drop "2, 16, repeat"
    w = w2
    put(2), put(2)                          # 'x', 'y'
    put(18), w.field(256 - 122 - 11, 7)     # zeros up to 255
    put(2), put(2)                          # 256, 257
    put(1)                                  # the distance code: 1 bit
    raw2 = bytearray()
    lit = {120: (0, 2), 121: (1, 2), 256: (2, 2), 257: (3, 2)}
    for ch in b"xyyx":
        w.code(*lit[ch])
        raw2.append(ch)
    w.code(*lit[257]), w.code(0, 1)          # length 3, distance 1
    raw2 += bytes([raw2[-1]]) * 3
    w.code(*lit[256])
    second = gzip_member(w.bytes(), bytes(raw2))
    return first + second, bytes(out) + bytes(raw2)


# ------------------------------------------------------------------------------- oracle
def test_codec_inputs_are_the_ones_that_were_compressed():
    for v in VECTORS:
        raw = ORIGINALS[v["name"]]
        assert len(raw) == v["size"] and hashlib.sha256(raw).hexdigest() == v["sha256"], v["name"]


@pytest.mark.parametrize("v", VECTORS, ids=lambda v: v["codec"] + "." + v["name"])
def test_oracle_decompresses_library_streams(orc, v):
    st, out, used = orc.decompress(v["codec"], base64.b64decode(v["compressed"]), v["size"])
    assert st == 0 and used == len(base64.b64decode(v["compressed"])) and bytes(out) == ORIGINALS[v["name"]]
    # the page header's size has to be what the stream holds
    for wrong in (v["size"] + 1, max(v["size"] - 1, 0)):
        if wrong != v["size"]:
            assert orc.decompress(v["codec"], base64.b64decode(v["compressed"]), wrong)[0] == orc.DECODE


@pytest.mark.parametrize("fx", PAGES, ids=lambda fx: fx["file"] + "." + fx["column"])
def test_oracle_decompresses_pyarrow_pages(orc, fx):
    import pyarrow as pa
    codec = PAGE_CODEC[fx["codec"]]
    for body, size in compressed_parts(fx):
        st, out, used = orc.decompress(codec, body, size)
        assert st == 0 and used == len(body)
        assert bytes(out) == (pa.Codec(codec.lower()).decompress(body, size, asbytes=True) if body else b"")


def zstd_block(kind, payload, size=None, last=False):
    """A raw (0) or RLE (1) block: 3-byte header, payload."""
    n = len(payload) if size is None else size
    return ((n << 3) | (kind << 1) | int(last)).to_bytes(3, "little") + payload


def handmade_zstd():
    """(stream, original): what the library's streams above do not hold: raw and RLE blocks, a frame without
    content size, several frames, a skippable frame, a content checksum."""
    import xxhash
    rng = np.random.default_rng(8)
    a, b = bytes(rng.integers(0, 256, 5000, dtype=np.uint8)), bytes(rng.integers(0, 256, 300, dtype=np.uint8))
    magic = (0xFD2FB528).to_bytes(4, "little")
    # frame 1: window descriptor, no content size; raw, RLE, empty raw, raw
    f1 = magic + bytes([0x00, 0x58]) + zstd_block(0, a) + zstd_block(1, b"\x07", 70000) + zstd_block(0, b"") + zstd_block(0, b, last=True)
    o1 = a + b"\x07" * 70000 + b
    skip = (0x184D2A53).to_bytes(4, "little") + (11).to_bytes(4, "little") + b"hello world"
    # frame 2: single segment, one-byte content size, content checksum
    o2 = b"checksummed " * 9
    f2 = magic + bytes([0x24, len(o2)]) + zstd_block(0, o2, last=True) + (xxhash.xxh64(o2).intdigest() & 0xFFFFFFFF).to_bytes(4, "little")
    # frame 3: a library frame (compressed blocks), its header rewritten to carry a checksum
    import pyarrow as pa
    o3 = (b"the quick brown fox jumps over the lazy dog. " * 400) + a + bytes(range(256)) * 40
    lib = pa.Codec("zstd", compression_level=3).compress(o3, asbytes=True)
    assert lib[:4] == magic and not lib[4] & 0x04
    f3 = lib[:4] + bytes([lib[4] | 0x04]) + lib[5:] + (xxhash.xxh64(o3).intdigest() & 0xFFFFFFFF).to_bytes(4, "little")
    return f1 + skip + f2 + f3, o1 + o2 + o3


def live_zstd_streams():
    """[(stream, original)] compressed now with the C library: inputs of several blocks at levels that choose
    other table modes (repeated tables, literals without a tree) than the committed vectors' levels do."""
    import pyarrow as pa
    rng = np.random.default_rng(9)
    words = [bytes(rng.integers(97, 123, int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(3000)]
    text = b" ".join(words[int(i)] for i in rng.zipf(1.3, 150000) % 3000)
    ints = np.cumsum(rng.integers(0, 50, 120000)).astype("<i8").tobytes()
    skew = bytes((rng.geometric(0.3, 400000) % 256).astype(np.uint8))
    out = []
    for raw in (text, ints, skew, text[:70000] + skew[:90000] + ints[:200000]):
        for level in (1, 3, 19):
            out.append((pa.Codec("zstd", compression_level=level).compress(raw, asbytes=True), raw))
    return out


def test_oracle_handmade_streams(orc):
    import pyarrow as pa
    for codec, (stream, raw) in (("SNAPPY", handmade_snappy()), ("SNAPPY", handmade_snappy_far_copies()),
                                 ("LZ4_RAW", handmade_lz4()), ("ZSTD", handmade_zstd()), ("GZIP", handmade_gzip())):
        st, out, used = orc.decompress(codec, stream, len(raw))
        assert st == 0 and bytes(out) == raw and used == len(stream)
        if codec == "GZIP":  # (members one after the other: Python's gzip module reads on, the one-shot call does not)
            import gzip
            import io
            assert gzip.GzipFile(fileobj=io.BytesIO(stream)).read() == raw
        elif codec != "ZSTD":  # (the library's one-shot call stops after the first frame)
            assert pa.Codec(codec.lower()).decompress(stream, len(raw), asbytes=True) == raw  # the C library reads them alike
    for stream, raw in live_zstd_streams():
        st, out, used = orc.decompress("ZSTD", stream, len(raw))
        assert st == 0 and bytes(out) == raw and used == len(stream)
    # broken streams
    D = orc.DECODE
    assert orc.decompress("SNAPPY", b"", 0)[0] == D                       # no preamble
    assert orc.decompress("SNAPPY", b"\x00", 0)[0] == 0                    # an empty page
    assert orc.decompress("SNAPPY", b"\x05\x10abcde", 5)[0] == 0
    assert orc.decompress("SNAPPY", b"\x05\x10abcd", 5)[0] == D            # literal runs past the stream
    assert orc.decompress("SNAPPY", b"\x06\x10abcde", 5)[0] == D           # preamble disagrees
    assert orc.decompress("SNAPPY", b"\x05\x00a\x01\x02", 5)[0] == D       # offset beyond what was produced
    assert orc.decompress("SNAPPY", b"\x05\x00a\x01\x00", 5)[0] == D       # offset 0
    assert orc.decompress("SNAPPY", b"\x05\x00a\x01\x01", 5)[:2][1].tobytes() == b"aaaaa"
    assert orc.decompress("SNAPPY", b"\x80\x80\x80\x80\x80\x01", 0)[0] == D  # preamble longer than a varint32
    assert orc.decompress("LZ4_RAW", b"", 0)[0] == 0
    assert orc.decompress("LZ4_RAW", b"\x10a", 1)[0] == 0
    assert orc.decompress("LZ4_RAW", b"\x10a", 2)[0] == D                  # short of the announced size
    assert orc.decompress("LZ4_RAW", b"\x20a", 2)[0] == D                  # literals run past the stream
    assert orc.decompress("LZ4_RAW", b"\x10a\x00\x00", 5)[0] == D          # offset 0
    assert orc.decompress("LZ4_RAW", b"\x10a\x02\x00", 5)[0] == D          # offset beyond what was produced
    assert orc.decompress("LZ4_RAW", b"\x10a\x01", 5)[0] == D              # half an offset
    assert orc.decompress("LZ4_RAW", b"\x1fa\x01\x00\xff", 300)[0] == D    # length bytes run out
    assert orc.decompress("LZ4_RAW", b"\x10a\x01\x00", 5)[:2][1].tobytes() == b"aaaaa"
    import zlib
    good = gzip_member(zlib.compress(b"hello hello hello", 9)[2:-4], b"hello hello hello")
    assert orc.decompress("GZIP", good, 17)[1].tobytes() == b"hello hello hello"
    assert orc.decompress("GZIP", b"", 0)[0] == 0                           # nothing is read from a page of no bytes
    assert orc.decompress("GZIP", b"", 1)[0] == D
    assert orc.decompress("GZIP", good, 16)[0] == D and orc.decompress("GZIP", good, 18)[0] == D  # the page header disagrees
    for at, byte in ((0, 0x1E), (1, 0x8A), (2, 7), (3, 0x20)):           # magic, method, a reserved flag
        assert orc.decompress("GZIP", good[:at] + bytes([byte]) + good[at + 1:], 17)[0] == D
    assert orc.decompress("GZIP", good[:-8] + bytes([good[-8] ^ 1]) + good[-7:], 17)[0] == D       # CRC-32
    assert orc.decompress("GZIP", good[:-4] + bytes([good[-4] ^ 1]) + good[-3:], 17)[0] == D       # ISIZE
    assert orc.decompress("GZIP", good[:-1], 17)[0] == D and orc.decompress("GZIP", good[:12], 17)[0] == D  # cut short
    assert orc.decompress("GZIP", good + b"\x1f", 17)[0] == D                # something that is no member behind the last one
    assert orc.decompress("GZIP", good + good, 34)[1].tobytes() == b"hello hello hello" * 2
    w = BitWriter()
    w.field(1, 1), w.field(3, 2)
    assert orc.decompress("GZIP", gzip_member(w.bytes(), b""), 0)[0] == D    # block type 3
    w = BitWriter()
    w.field(1, 1), w.field(0, 2), w.raw((3).to_bytes(2, "little") + (0xFFFD).to_bytes(2, "little") + b"abc")
    assert orc.decompress("GZIP", gzip_member(w.bytes(), b"abc"), 3)[0] == D  # LEN and NLEN disagree
    w = BitWriter()
    w.field(1, 1), w.field(1, 2), fixed_literal(w, ord("a")), fixed_literal(w, 257), w.code(1, 5), fixed_literal(w, 256)
    assert orc.decompress("GZIP", gzip_member(w.bytes(), b"aaaa"), 4)[0] == D  # distance 2 behind one byte
    w = BitWriter()
    w.field(1, 1), w.field(1, 2), fixed_literal(w, ord("a")), fixed_literal(w, 257), w.code(30, 5), fixed_literal(w, 256)
    assert orc.decompress("GZIP", gzip_member(w.bytes(), b"aaaa"), 4)[0] == D  # distance symbol 30
    w = BitWriter()
    w.field(1, 1), w.field(1, 2), fixed_literal(w, ord("a")), fixed_literal(w, 286), w.code(0, 5), fixed_literal(w, 256)
    assert orc.decompress("GZIP", gzip_member(w.bytes(), b"aaaa"), 4)[0] == D  # length symbol 286
    w = BitWriter()
    w.field(1, 1), w.field(2, 2), w.field(30, 5), w.field(0, 5), w.field(0, 4)
    assert orc.decompress("GZIP", gzip_member(w.bytes() + bytes(8), b""), 0)[0] == D  # 287 literal / length codes
    magic = (0xFD2FB528).to_bytes(4, "little")
    assert orc.decompress("ZSTD", b"", 0)[0] == 0                          # no frame at all: an empty page
    assert orc.decompress("ZSTD", magic + b"\x20\x00" + zstd_block(0, b"", last=True), 0)[0] == 0
    assert orc.decompress("ZSTD", magic + b"\x20\x03" + zstd_block(0, b"abc", last=True), 3)[1].tobytes() == b"abc"
    assert orc.decompress("ZSTD", magic + b"\x20\x04" + zstd_block(0, b"abc", last=True), 3)[0] == D  # content size disagrees
    assert orc.decompress("ZSTD", magic + b"\x20\x03" + zstd_block(0, b"abc", last=True), 4)[0] == D  # page header disagrees
    assert orc.decompress("ZSTD", magic + b"\x20\x03" + zstd_block(0, b"abc"), 3)[0] == D             # no last block
    assert orc.decompress("ZSTD", magic + b"\x21\x01\x03" + zstd_block(0, b"abc", last=True), 3)[0] == D  # wants a dictionary
    assert orc.decompress("ZSTD", magic + b"\x28\x03" + zstd_block(0, b"abc", last=True), 3)[0] == D  # reserved bit
    assert orc.decompress("ZSTD", magic + b"\x20\x03" + zstd_block(3, b"abc", last=True), 3)[0] == D  # reserved block type
    assert orc.decompress("ZSTD", magic + b"\x24\x03" + zstd_block(0, b"abc", last=True) + b"\0\0\0\0", 3)[0] == D  # wrong checksum
    assert orc.decompress("ZSTD", b"\x28\xb5\x2f\xfc\x20\x00\x01\x00\x00", 0)[0] == D            # wrong magic


# ------------------------------------------------------------------------------- GPU
def run_batch(pq, gh, torch, codec, items, misalign=3):
    """items: (stream, announced size); one pqd_decompress_pages_dev call.  -> [(status, bytes, consumed)]"""
    n = len(items)
    descs = (pq.DecompressDesc * n)()
    src_at, pos = [], misalign
    for stream, size in items:
        src_at.append(pos)
        pos += len(stream) + 1  # back to back at odd addresses
    blob = np.full(pos + 64, 0xEE, np.uint8)
    for (stream, size), a in zip(items, src_at):
        blob[a:a + len(stream)] = np.frombuffer(bytes(stream), np.uint8)
    d_blob = gh.to_dev(blob)
    dst_at, pos = [], 5
    for stream, size in items:
        dst_at.append(pos)
        pos += size + 7
    d_out = torch.full((pos + 64,), 0xA5, dtype=torch.uint8, device=gh.dev())
    for i, (stream, size) in enumerate(items):
        descs[i].src, descs[i].dst = d_blob.data_ptr() + src_at[i], d_out.data_ptr() + dst_at[i]
        descs[i].src_len, descs[i].dst_len, descs[i].codec = len(stream), size, codec
    d_descs = gh.to_dev(np.frombuffer(bytes(descs), np.uint8).copy())
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=gh.dev())
    pq.check(pq.lib.pqd_decompress_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), n, codec, None))
    torch.cuda.synchronize()
    raw = d_res.cpu().numpy().view(np.uint32).reshape(-1, 4)
    out = d_out.cpu().numpy()
    res = []
    for i, (stream, size) in enumerate(items):
        st, made, used = int(raw[i, 0]), int(raw[i, 1]), int(raw[i, 2])
        assert made <= size
        assert (out[dst_at[i] + size:dst_at[i] + size + 7] == 0xA5).all() and (out[dst_at[i] - 5:dst_at[i]] == 0xA5).all(), \
            "bytes written outside the page"
        res.append((st, out[dst_at[i]:dst_at[i] + made].tobytes(), used))
    return res


def check_batch(orc, codec_name, items, results):
    for (stream, size), (st, data, used) in zip(items, results):
        ost, odata, oused = orc.decompress(codec_name, stream, size)
        assert st == ost, (st, ost, len(stream), size)
        assert data == bytes(odata)
        assert used == oused


@pytest.mark.gpu
@pytest.mark.parametrize("codec_name", CODECS)
def test_gpu_decompresses_library_streams_and_pages(orc, codec_name):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    items, want = [], []
    for v in VECTORS:
        if v["codec"] == codec_name:
            items.append((base64.b64decode(v["compressed"]), v["size"]))
            want.append(ORIGINALS[v["name"]])
    stream, raw = {"SNAPPY": handmade_snappy, "LZ4_RAW": handmade_lz4, "ZSTD": handmade_zstd, "GZIP": handmade_gzip}[codec_name]()
    items.append((stream, len(raw)))
    want.append(raw)
    if codec_name == "ZSTD":
        magic = (0xFD2FB528).to_bytes(4, "little")
        for stream, raw in live_zstd_streams() + [
                (b"", b""), (magic + b"\x20\x00" + zstd_block(0, b"", last=True), b""),
                (magic + b"\x20\x03" + zstd_block(0, b"abc", last=True), b"abc")]:
            items.append((stream, len(raw)))
            want.append(raw)
    for fx in PAGES:
        if PAGE_CODEC[fx["codec"]] == codec_name:
            for body, size in compressed_parts(fx):
                items.append((body, size))
                want.append(None)
    for misalign in (0, 3):
        results = run_batch(pq, gh, torch, CODEC_NUM[codec_name], items, misalign)
        check_batch(orc, codec_name, items, results)
        for (st, data, used), w in zip(results, want):
            assert st == 0 and (w is None or data == w)
    # announced sizes that do not fit the stream
    wrong = [(s, size + 1) for s, size in items[:8]] + [(s, size - 1) for s, size in items[:8] if size]
    check_batch(orc, codec_name, wrong, run_batch(pq, gh, torch, CODEC_NUM[codec_name], wrong))


@pytest.mark.gpu
def test_gpu_lz4_stream_that_ends_with_a_lone_token(orc):
    """The last sequence of an LZ4 block has no match; with no literals either it is one byte long, the only sequence of
    fewer than two bytes in either format.  Streams that end that way behind a match, at every place of a 32-byte
    segment, and the one-byte stream of an empty page; a batch of ordinary pages runs in front, so that the tables in
    LDS hold another page's entries."""
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(77)
    filler = []
    for v in VECTORS:
        if v["codec"] == "LZ4_RAW":
            filler.append((base64.b64decode(v["compressed"]), v["size"]))
    items = [(b"\x00", 0)]
    for n_lit in range(1, 40):
        lits = bytes(rng.integers(97, 123, n_lit, dtype=np.uint8))
        head = bytes([0xF0, n_lit - 15]) if n_lit >= 15 else bytes([n_lit << 4])
        items.append((head + lits + b"\x01\x00" + b"\x00", n_lit + 4))  # literals, a match of 4 from 1 back, a token alone
    for stream, size in items:
        assert orc.decompress("LZ4_RAW", stream, size)[0] == 0
    for misalign in (0, 3):
        run_batch(pq, gh, torch, CODEC_NUM["LZ4_RAW"], filler, misalign)
        results = run_batch(pq, gh, torch, CODEC_NUM["LZ4_RAW"], items, misalign)
        check_batch(orc, "LZ4_RAW", items, results)
        assert all(st == 0 for st, _, _ in results)


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["staged8", "pipe4", "pipe16", "serial4", "serial16"])
@pytest.mark.parametrize("codec_name", ["SNAPPY", "LZ4_RAW"])
def test_gpu_lz_pages_with_other_kernels_and_waves(orc, monkeypatch, codec_name, kernel):
    """Calls of few pages (every test here) take the Snappy / LZ4 kernel whose first four waves make the two halves of
    a window's table, follow the chain through the window before and list the elements of the one before that while
    twelve others execute a fifth; calls of up to 512 pages take it with eight waves (the table by one wave), up to 768
    the kernel whose first wave parses while three others execute, larger ones the kernel whose four waves parse, then
    execute: the same streams with kernel and waves fixed."""
    monkeypatch.setenv("PQD_LZ_PIPE", {"serial": "0", "pipe": "1", "staged": "2"}[kernel.rstrip("0123456789")])
    monkeypatch.setenv("PQD_LZ_WAVES", kernel.lstrip("pipeserialstaged"))
    test_gpu_decompresses_library_streams_and_pages(orc, codec_name)
    test_gpu_damaged_streams(orc, codec_name)
    if codec_name == "LZ4_RAW":
        test_gpu_lz4_stream_that_ends_with_a_lone_token(orc)
    else:
        test_gpu_snappy_far_copies_in_front_of_literals(orc)


@pytest.mark.gpu
def test_gpu_zstd_pages_with_one_wave(orc, monkeypatch):
    """The ZSTD kernel gives a page two waves (the first decodes, the second moves the bytes: every test here); the form
    with one wave per page is kept for comparison: the same streams with it."""
    monkeypatch.setenv("PQD_ZSTD_WAVES", "1")
    test_gpu_decompresses_library_streams_and_pages(orc, "ZSTD")
    test_gpu_damaged_streams(orc, "ZSTD")


@pytest.mark.gpu
def test_gpu_gzip_pages_with_the_small_ring(orc, monkeypatch):
    """The GZIP kernel keeps 16 KiB of a page's output in LDS in calls whose pages are all in flight with that (every
    test here), 8 KiB in larger ones (matches from farther back are read from memory): the same streams with 8 KiB."""
    import gzip
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    monkeypatch.setenv("PQD_GZIP_RING", "8")
    test_gpu_decompresses_library_streams_and_pages(orc, "GZIP")
    test_gpu_damaged_streams(orc, "GZIP")
    # matches from 10 to 30 KB back: behind the small ring, inside the format's 32 KiB
    rng = np.random.default_rng(5)
    items, raws = [], []
    for gap in (10_000, 20_000, 30_000):
        block = rng.integers(0, 256, gap, dtype=np.uint8).tobytes()
        raw = block + block[:7000] + b"abc" * 100 + block[3000:9000] + block
        raws.append(raw)
        items.append((gzip.compress(raw, compresslevel=9, mtime=0), len(raw)))
    for env in ("8", "16"):
        monkeypatch.setenv("PQD_GZIP_RING", env)
        results = run_batch(pq, gh, torch, CODEC_NUM["GZIP"], items)
        for (st, data, used), raw, (stream, size) in zip(results, raws, items):
            assert st == 0 and data == raw and used == len(stream)
        check_batch(orc, "GZIP", items, results)


@pytest.mark.gpu
def test_gpu_snappy_far_copies_in_front_of_literals(orc):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    stream, raw = handmade_snappy_far_copies()
    assert orc.decompress("SNAPPY", stream, len(raw))[1].tobytes() == raw
    items = [(stream, len(raw))] * 3
    for misalign in (0, 5):
        for st, data, used in run_batch(pq, gh, torch, CODEC_NUM["SNAPPY"], items, misalign):
            assert st == 0 and used == len(stream)
            assert data == raw


@pytest.mark.gpu
def test_gpu_descriptor_of_another_codec_is_rejected(orc):
    """A batch is one codec plus stored parts: a descriptor that names another codec gets PQD_BAD_ARG, not what
    the result array held before."""
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    snappy, raw = handmade_snappy()
    lz4, raw4 = handmade_lz4()
    n = 3
    descs = (pq.DecompressDesc * n)()
    streams = [(snappy, len(raw), CODEC_NUM["SNAPPY"]), (lz4, len(raw4), CODEC_NUM["LZ4_RAW"]), (b"stored", 6, 0)]
    keep = []
    for i, (stream, size, codec) in enumerate(streams):
        src, dst = gh.to_dev(np.frombuffer(stream, np.uint8)), torch.zeros(size + 16, dtype=torch.uint8, device=gh.dev())
        keep += [src, dst]
        descs[i].src, descs[i].dst, descs[i].src_len, descs[i].dst_len, descs[i].codec = src.data_ptr(), dst.data_ptr(), len(stream), size, codec
    d_descs = gh.to_dev(np.frombuffer(bytes(descs), np.uint8).copy())
    for call_codec, want in ((CODEC_NUM["SNAPPY"], [0, pq.BAD_ARG, 0]), (CODEC_NUM["LZ4_RAW"], [pq.BAD_ARG, 0, 0]),
                             (0, [pq.BAD_ARG, pq.BAD_ARG, 0]), (CODEC_NUM["ZSTD"], [pq.BAD_ARG, pq.BAD_ARG, 0])):
        d_res = torch.full((n * 16,), 0x77, dtype=torch.uint8, device=gh.dev())
        pq.check(pq.lib.pqd_decompress_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), n, call_codec, None))
        torch.cuda.synchronize()
        assert d_res.cpu().numpy().view(np.int32).reshape(-1, 4)[:, 0].tolist() == want


@pytest.mark.gpu
@pytest.mark.parametrize("codec_name", CODECS)
def test_gpu_damaged_streams(orc, codec_name):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(123)
    base = [(base64.b64decode(v["compressed"]), v["size"]) for v in VECTORS
            if v["codec"] == codec_name and base_name(v["name"]) in ("short", "text", "ints", "mixed", "run_3", "timestamps")]
    if codec_name == "ZSTD":
        magic = (0xFD2FB528).to_bytes(4, "little")
        base += [(s, len(r)) for s, r in live_zstd_streams()[:6]] + [(handmade_zstd()[0], len(handmade_zstd()[1]))]
        heads = [magic + b"\x20\x04" + zstd_block(0, b"abc", last=True), magic + b"\x20\x03" + zstd_block(0, b"abc"),
                 magic + b"\x21\x01\x03" + zstd_block(0, b"abc", last=True), magic + b"\x28\x03" + zstd_block(0, b"abc", last=True),
                 magic + b"\x20\x03" + zstd_block(3, b"abc", last=True),
                 magic + b"\x24\x03" + zstd_block(0, b"abc", last=True) + b"\0\0\0\0", b"\x28\xb5\x2f\xfc\x20\x00\x01\x00\x00"]
        check_batch(orc, codec_name, [(h, 3) for h in heads], run_batch(pq, gh, torch, CODEC_NUM[codec_name], [(h, 3) for h in heads]))
    items = []
    for trial in range(150):
        stream, size = base[trial % len(base)]
        stream = bytearray(stream[:4000]) if trial % 2 else bytearray(stream)
        for _ in range(int(rng.integers(1, 4))):
            stream[int(rng.integers(0, min(len(stream), 600)))] = int(rng.integers(0, 256))
        if trial % 7 == 0:
            stream = stream[:int(rng.integers(0, len(stream)))]
        items.append((bytes(stream), size))
    check_batch(orc, codec_name, items, run_batch(pq, gh, torch, CODEC_NUM[codec_name], items))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["staged16", "staged8", "pipe4", "serial4"])
@pytest.mark.parametrize("codec_name", ["SNAPPY", "LZ4_RAW"])
def test_gpu_lz_long_streams_damaged_anywhere(orc, monkeypatch, codec_name, kernel):
    """Streams of a hundred windows and more (text, bytes that do not compress: literals that reach over many windows;
    zeros and a short period: matches of many length bytes; small integers), whole and damaged anywhere, cut short, and
    with a wrong size announced: the windows the kernels have in flight behind a sequence that breaks a rule must not
    count.  Status, bytes and position as the oracle's, for every kind of kernel."""
    pa = pytest.importorskip("pyarrow")
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    monkeypatch.setenv("PQD_LZ_PIPE", {"serial": "0", "pipe": "1", "staged": "2"}[kernel.rstrip("0123456789")])
    monkeypatch.setenv("PQD_LZ_WAVES", kernel.lstrip("pipeserialstaged"))
    rng = np.random.default_rng(77)
    words = [bytes(rng.integers(97, 123, int(k), dtype=np.uint8)) for k in rng.integers(2, 10, 500)]

    def content(seed):
        r = np.random.default_rng(seed)
        parts = []
        for kind in r.permutation(6):
            if kind == 0:
                parts.append(b" ".join(words[i] for i in r.integers(0, 500, 8000)))
            elif kind == 1:
                parts.append(r.integers(0, 256, int(r.integers(3000, 30000)), dtype=np.uint8).tobytes())
            elif kind == 2:
                parts.append(bytes(int(r.integers(1000, 70000))))
            elif kind == 3:
                parts.append(bytes(r.integers(0, 256, int(r.integers(1, 9)), dtype=np.uint8)) * int(r.integers(500, 6000)))
            elif kind == 4:
                parts.append(r.integers(0, 1000, 12000).astype("<i4").tobytes())
            else:
                parts.append(np.round(r.normal(size=6000), 2).astype("<f8").tobytes())
        return b"".join(parts)
    codec = pa.Codec({"SNAPPY": "snappy", "LZ4_RAW": "lz4_raw"}[codec_name])
    raws = [content(s) for s in range(6)]
    streams = [codec.compress(r, asbytes=True) for r in raws]
    items = [(st, len(r)) for st, r in zip(streams, raws)]
    for trial in range(54):
        stream, size = bytearray(streams[trial % 6]), len(raws[trial % 6])
        for _ in range(int(rng.integers(1, 3))):
            stream[int(rng.integers(0, len(stream)))] = int(rng.integers(0, 256))
        if trial % 5 == 0:
            stream = stream[:int(rng.integers(len(stream) // 2, len(stream)))]
        if trial % 9 == 0:
            size = max(0, size + int(rng.integers(-5000, 5000)))
        items.append((bytes(stream), size))
    results = run_batch(pq, gh, torch, CODEC_NUM[codec_name], items)
    for k in range(6):
        assert results[k][0] == 0 and results[k][1] == raws[k]
    check_batch(orc, codec_name, items, results)


@pytest.mark.gpu
@pytest.mark.parametrize("codec_name,n_pages", [(c, n) for c in ("SNAPPY", "LZ4_RAW", "ZSTD") for n in (300, 600, 900)] + [("GZIP", 1400)])
def test_gpu_calls_of_many_pages_take_their_own_kernels(orc, codec_name, n_pages):
    """The kernel (its waves per page, the bytes of output it keeps in LDS) follows from the number of pages in the call:
    calls of 300, 600 and 900 pages (GZIP: 1400), whole and damaged streams mixed, every result as the oracle's."""
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(n_pages)
    base = [(base64.b64decode(v["compressed"]), v["size"]) for v in VECTORS if v["codec"] == codec_name]
    items = []
    for i in range(n_pages):
        stream, size = base[i % len(base)]
        if i % 11 == 3:
            stream = bytearray(stream)
            stream[int(rng.integers(0, len(stream)))] ^= 1 << int(rng.integers(0, 8))
            stream = bytes(stream)
        items.append((stream, size))
    results = run_batch(pq, gh, torch, CODEC_NUM[codec_name], items)
    expected = {}
    for (stream, size), (st, data, used) in zip(items, results):
        key = (stream, size)
        if key not in expected:
            ost, odata, oused = orc.decompress(codec_name, stream, size)
            expected[key] = (ost, bytes(odata), oused)
        assert (st, data, used) == expected[key]


@pytest.mark.gpu
def test_gpu_stored_parts_and_host_entry(orc):
    import torch
    import parquet_java_amd as pq
    import gpu_harness as gh
    pq.init(0)
    rng = np.random.default_rng(7)
    items = [(bytes(rng.integers(0, 256, n, dtype=np.uint8)), n) for n in (0, 1, 15, 16, 17, 1000, 70001)]
    for (stream, size), (st, data, used) in zip(items, run_batch(pq, gh, torch, 0, items)):
        assert st == 0 and data == stream and used == size
    (st, data, used), = run_batch(pq, gh, torch, 0, [(b"abcdef", 4)])
    assert st == orc.DECODE and data == b"abcd"
    assert pq.lib.pqd_codec_supported(1) and pq.lib.pqd_codec_supported(7) and pq.lib.pqd_codec_supported(6)
    assert pq.lib.pqd_codec_supported(2) and not pq.lib.pqd_codec_supported(3) and not pq.lib.pqd_codec_supported(4)
    # the host entry behind the BytesDecompressor mirror
    for name in CODECS:
        d = pq.BytesDecompressor(name)
        for v in VECTORS:
            if v["codec"] == name and base_name(v["name"]) in ("empty", "short", "text", "random"):
                assert d.decompress(base64.b64decode(v["compressed"]), v["size"]) == ORIGINALS[v["name"]]
        with pytest.raises(IOError):
            d.decompress(b"\x05\x10abcd", 5)
    with pytest.raises(pq.IllegalArgumentException):
        pq.BytesDecompressor("BROTLI")
