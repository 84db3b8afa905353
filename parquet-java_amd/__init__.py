"""parquet page decode on MI355X -- host-side mirror of parquet-java's reader interface.

The classes here keep the names, argument meaning and failure behaviour of the reference's
``ParquetReadRouter`` / ``ValuesReader`` surface and forward every decode to the HIP library through
its C ABI (``include/pqd.h``).  They exist so callers (and the parity tests) can be written the way
the reference's own callers are; the compute is in ``csrc/*.hip``.  There is no CPU decode path in
this package.

Reference (paths relative to the parquet-java tree):
  parquet-plugins/parquet-encoding-vector/.../ParquetReadRouter.java:57-116
  parquet-column/.../column/values/ValuesReader.java:36-203
  parquet-common/.../bytes/ByteBufferInputStream.java:35-151
"""
import ctypes as _C

import numpy as _np

from . import _native
from ._native import (BAD_ARG, CODEC_GZIP, CODEC_LZ4_RAW, CODEC_SNAPPY, CODEC_UNCOMPRESSED, CODEC_ZSTD, DECODE, DICT_OOB, ENC_BIT_PACKED, ENC_BYTE_STREAM_SPLIT,  # noqa: F401
                      ENC_DELTA_BINARY_PACKED, ENC_DELTA_BYTE_ARRAY, ENC_DELTA_LENGTH_BYTE_ARRAY, ENC_PLAIN, ENC_PLAIN_BOOLEAN, ENC_PLAIN_BYTE_ARRAY, ENC_RLE, ENC_RLE_DICTIONARY, EOF, F_LENGTH_PREFIX, F_WIDTH_PREFIX,
                      HIP_ERROR, NO_DEVICE, OK, PAGE_DATA, PAGE_DATA_V2, PAGE_DICTIONARY, PAGE_INDEX, PAST_END, TYPE_BOOLEAN, TYPE_BYTE_ARRAY, TYPE_DOUBLE, TYPE_FIXED_LEN_BYTE_ARRAY, TYPE_FLOAT,
                      TYPE_INT32, TYPE_INT64, TYPE_INT96, VIEW_INVALID,
                      BinaryPageDesc, ChunkOutputs, ChunkPage, ColumnMeta, DecompressDesc, DeltaBytesDesc, DictInfo, OffsetsDesc, PageDesc, PageResult, PqdError, ScatterDesc, ValidityDesc, check, lib)

__all__ = ["ParquetReadRouter", "Packer", "BytePacker", "BytePackerForLong", "ByteBufferInputStream", "ValuesReader", "RunLengthBitPackingHybridValuesReader",
           "DictionaryValuesReader", "PlainBinaryDictionary", "BytesDecompressor", "DeltaBinaryPackingValuesReader", "DeltaLengthByteArrayValuesReader", "DeltaByteArrayReader", "BinaryPlainValuesReader", "ByteStreamSplitValuesReader", "PlainValuesReader", "BooleanPlainValuesReader",
           "ByteBitPackingValuesReader", "ZeroIntegerValuesReader", "ParquetDecodingException", "EOFException",
           "IllegalArgumentException", "device_count", "init", "decode_pages_dev", "make_descs", "lib"]


# ------------------------------------------------------------------------------- exceptions
class ParquetDecodingException(RuntimeError):
    """org.apache.parquet.io.ParquetDecodingException"""


class EOFException(EOFError):
    """java.io.EOFException"""


class IllegalArgumentException(ValueError):
    """java.lang.IllegalArgumentException"""


def _raise_for(status, what=""):
    if status == OK:
        return
    if status == EOF:
        raise EOFException(what)
    if status == BAD_ARG:
        raise IllegalArgumentException(what or lib.pqd_last_error().decode())
    if status == PAST_END:
        raise IllegalArgumentException("Reading past RLE/BitPacking stream.")
    if status in (DECODE, DICT_OOB):
        raise ParquetDecodingException(what or lib.pqd_status_string(status).decode())
    raise PqdError(status, what)


def device_count():
    return lib.pqd_device_count()


def init(device=0):
    check(lib.pqd_init(device), "pqd_init")


# ------------------------------------------------------------------------------- streams
class ByteBufferInputStream:
    """The slice of ByteBufferInputStream the decoders use (single-buffer flavour:
    parquet-common/.../bytes/SingleBufferInputStream.java:43-177)."""

    def __init__(self, buf, start=0, end=None):
        self._buf = _np.frombuffer(bytes(buf), dtype=_np.uint8) if not isinstance(buf, _np.ndarray) else buf
        self._start = start
        self._pos = start
        self._end = self._buf.size if end is None else end

    @staticmethod
    def wrap(buf):
        return ByteBufferInputStream(buf)

    def position(self):
        return self._pos - self._start

    def available(self):
        return self._end - self._pos

    def read(self):
        if self._pos >= self._end:
            raise EOFException()
        b = int(self._buf[self._pos])
        self._pos += 1
        return b

    def skipFully(self, n):
        if n > self.available():
            raise EOFException()
        self._pos += n

    def slice(self, length):
        if self.available() < length:
            raise EOFException()
        out = self._buf[self._pos:self._pos + length]
        self._pos += length
        return out

    def sliceStream(self, length):
        if self.available() < length:
            raise EOFException()
        s = ByteBufferInputStream(self._buf, self._pos, self._pos + length)
        self._pos += length
        return s

    def remainingStream(self):
        return self.sliceStream(self.available())

    def _peek_all(self):
        return self._buf[self._pos:self._end]


def _ptr(a):
    return a.ctypes.data_as(_C.c_void_p) if a.size else None


# ------------------------------------------------------------------------------- ParquetReadRouter
class ParquetReadRouter:
    """Bulk unpack of one bit-packed run into ``currentBuffer[0:currentCount]``.

    Same contract as ParquetReadRouter.read / readBatchUsing512Vector / readBatch
    (ParquetReadRouter.java:57-116): consumes exactly ``currentCount*bitWidth/8`` bytes of ``in``,
    EOFException if fewer remain.  All three names run the device kernel (``pqd_unpack_i32``).
    """

    @staticmethod
    def read(bitWidth, in_, currentCount, currentBuffer):
        total = currentCount * bitWidth // 8
        buf = _np.ascontiguousarray(in_.slice(total))  # EOFException if short, as the Java slice()
        out = currentBuffer
        if not (isinstance(out, _np.ndarray) and out.dtype == _np.int32 and out.flags.c_contiguous):
            raise IllegalArgumentException("currentBuffer must be a contiguous int32 array")
        if out.size < currentCount:
            raise IndexError("currentBuffer too small")  # ArrayIndexOutOfBoundsException
        st = lib.pqd_unpack_i32(bitWidth, _ptr(buf), buf.size, _ptr(out), currentCount)
        _raise_for(st, "ParquetReadRouter.read")

    readBatchUsing512Vector = read
    readBatch = read

    @staticmethod
    def readLongs(bitWidth, in_, currentCount, currentBuffer):
        """BytePackerForLong.unpack8Values over a whole run (BytePackerForLong.java:72-105)."""
        total = currentCount // 8 * bitWidth
        buf = _np.ascontiguousarray(in_.slice(total))
        st = lib.pqd_unpack_i64(bitWidth, _ptr(buf), buf.size, _ptr(currentBuffer), currentCount)
        _raise_for(st, "ParquetReadRouter.readLongs")


# ------------------------------------------------------------------------------- BytePacker / Packer
class BytePacker:
    """Per-step unpackers of one bit width: BytePacker.unpack8Values / unpack32Values /
    unpackValuesUsingVector / getUnpackCount (BytePacker.java:40-148), as generated into
    ByteBitPackingLE/BE.Packer0..32 (ByteBasedBitPackingGenerator.java:258-308) and implemented by the
    vector plugin (ByteBitPacking512VectorLE.java:165-3178).  Every call is one device call
    (``pqd_unpack_i32`` for LSB-first, a BIT_PACKED page for MSB-first): the compatibility path for
    callers that are written against this interface; bulk callers use ``ParquetReadRouter``."""

    def __init__(self, bitWidth, msb_first=False):
        if not 0 <= bitWidth <= 32:
            raise IllegalArgumentException("bitWidth must be in 0..32")
        self._w, self._msb = bitWidth, msb_first

    def getBitWidth(self):
        return self._w

    def getUnpackCount(self):
        # values per unpackValuesUsingVector step (ByteBitPacking512VectorLE: 32 for widths 1-7 and
        # 9-15, 64 for width 8, 16 for widths 16-32, 0 for width 0)
        w = self._w
        return 0 if w == 0 else 64 if w == 8 else 32 if w < 16 else 16

    def _unpack(self, n, input, inPos, output, outPos):
        if not (isinstance(output, _np.ndarray) and output.dtype == _np.int32 and output.flags.c_contiguous):
            raise IllegalArgumentException("output must be a contiguous int32 array")
        if outPos < 0 or outPos + n > output.size:
            raise IndexError("output too small")  # ArrayIndexOutOfBoundsException
        nbytes = n * self._w // 8
        buf = _np.frombuffer(input, _np.uint8) if not isinstance(input, _np.ndarray) else input.view(_np.uint8).reshape(-1)
        if inPos < 0 or inPos + nbytes > buf.size:
            raise IndexError("input too short")  # IndexOutOfBoundsException
        if n == 0 or self._w == 0:
            output[outPos:outPos + n] = 0 if n else output[outPos:outPos]
            return
        src = _np.ascontiguousarray(buf[inPos:inPos + nbytes])
        dst = output[outPos:outPos + n]
        if not self._msb:
            _raise_for(lib.pqd_unpack_i32(self._w, _ptr(src), src.size, _ptr(dst), n), "BytePacker.unpack")
            return
        d = PageDesc()
        d.data, d.data_len, d.out, d.num_values = src.ctypes.data, src.size, dst.ctypes.data, n
        d.encoding, d.elem_size, d.bit_width = ENC_BIT_PACKED, 4, self._w
        r = PageResult()
        check(lib.pqd_decode_page(_C.byref(d), _C.byref(r)), "pqd_decode_page")
        _raise_for(r.status, "BytePacker.unpack")

    def unpack8Values(self, input, inPos, output, outPos):
        self._unpack(8, input, inPos, output, outPos)

    def unpack32Values(self, input, inPos, output, outPos):
        self._unpack(32, input, inPos, output, outPos)

    def unpackValuesUsingVector(self, input, inPos, output, outPos):
        if self._msb:
            raise RuntimeError("unpackValuesUsingVector must be implemented by subclass!")  # BytePacker.java:139-147
        self._unpack(self.getUnpackCount(), input, inPos, output, outPos)

    def pack8Values(self, *a):
        raise RuntimeError("pack is not part of the decode library")  # the vector plugin throws too (:3229-3232)

    pack32Values = pack8Values


class BytePackerForLong:
    """BytePackerForLong.unpack8Values / unpack32Values (BytePackerForLong.java:72-105), widths 0..64,
    LSB-first (the delta reader's unpacker)."""

    def __init__(self, bitWidth):
        if not 0 <= bitWidth <= 64:
            raise IllegalArgumentException("bitWidth must be in 0..64")
        self._w = bitWidth

    def getBitWidth(self):
        return self._w

    def _unpack(self, n, input, inPos, output, outPos):
        if not (isinstance(output, _np.ndarray) and output.dtype == _np.int64 and output.flags.c_contiguous):
            raise IllegalArgumentException("output must be a contiguous int64 array")
        if outPos < 0 or outPos + n > output.size:
            raise IndexError("output too small")
        nbytes = n * self._w // 8
        buf = _np.frombuffer(input, _np.uint8) if not isinstance(input, _np.ndarray) else input.view(_np.uint8).reshape(-1)
        if inPos < 0 or inPos + nbytes > buf.size:
            raise IndexError("input too short")
        src = _np.ascontiguousarray(buf[inPos:inPos + nbytes])
        dst = output[outPos:outPos + n]
        _raise_for(lib.pqd_unpack_i64(self._w, _ptr(src), src.size, _ptr(dst), n), "BytePackerForLong.unpack")

    def unpack8Values(self, input, inPos, output, outPos):
        self._unpack(8, input, inPos, output, outPos)

    def unpack32Values(self, input, inPos, output, outPos):
        self._unpack(32, input, inPos, output, outPos)


class Packer:
    """Packer.LITTLE_ENDIAN / Packer.BIG_ENDIAN factories (Packer.java:26-121).  ``newBytePackerVector``
    is the plugin hook (Packer.java:92-111): here the device library is the plugin."""

    def __init__(self, msb_first):
        self._msb = msb_first

    def newBytePacker(self, width):
        return BytePacker(width, self._msb)

    def newBytePackerVector(self, width):
        if self._msb:
            raise RuntimeError("Not currently supported!")  # Packer.java:61-64
        if device_count() < 1:
            raise RuntimeError("No enable java vector plugin on little endian architectures")  # Packer.java:104-108
        return BytePacker(width, False)

    def newBytePackerForLong(self, width):
        if self._msb:
            raise RuntimeError("MSB-first 64-bit packers are not on the decode path")
        return BytePackerForLong(width)


Packer.LITTLE_ENDIAN = Packer(False)
Packer.BIG_ENDIAN = Packer(True)


# ------------------------------------------------------------------------------- ValuesReader family
class ValuesReader:
    """Eager page reader: ``initFromPage`` decodes the page on the device (``pqd_decode_page``),
    ``read*()`` serve from the decoded array, failures surface at the read that would have
    raised them in the reference (ValuesReader.java:107-202)."""

    _encoding = None
    _dtype = _np.int32
    _flags = 0
    _bit_width = 0

    def __init__(self):
        self._values = _np.zeros(0, self._dtype)
        self._status = OK
        self._i = 0
        self._next_offset = None

    # -- the part subclasses parameterise
    def _dict_array(self):
        return None

    def _elem_size(self):
        return _np.dtype(self._dtype).itemsize

    def initFromPage(self, valueCount, stream):
        start = stream.position()
        data = _np.ascontiguousarray(stream._peek_all())
        out = _np.zeros(max(valueCount, 1) * self._elem_size(), _np.uint8)
        d = PageDesc()
        d.data = data.ctypes.data if data.size else None
        d.data_len = data.size
        d.out = out.ctypes.data
        d.num_values = valueCount
        d.encoding = self._encoding
        d.elem_size = self._elem_size()
        d.bit_width = self._bit_width
        d.flags = self._flags
        dic = self._dict_array()
        if dic is not None:
            d.dict = dic.ctypes.data if dic.size else None
            d.dict_len = dic.size
        r = PageResult()
        check(lib.pqd_decode_page(_C.byref(d), _C.byref(r)), "pqd_decode_page")
        self._status = r.status
        self._values = out[:r.values_written * self._elem_size()].view(self._dtype) if self._elem_size() in (1, 2, 4, 8) \
            else out[:r.values_written * self._elem_size()].reshape(-1, self._elem_size())
        self._i = 0
        self._consume(stream, r, start)

    def _consume(self, stream, r, start):
        stream.skipFully(stream.available())  # hybrid readers take the rest of the page

    def getNextOffset(self):
        if self._next_offset is None:
            raise ParquetDecodingException("Unsupported: cannot get offset of the next section.")
        return self._next_offset

    def _next(self):
        if self._i >= len(self._values):
            self._fail_at_end()
        v = self._values[self._i]
        self._i += 1
        return v

    def _fail_at_end(self):
        _raise_for(self._status if self._status != OK else PAST_END)

    def readInteger(self):
        return int(self._next())

    def readLong(self):
        return int(self._next())

    def readFloat(self):
        return float(self._next())

    def readDouble(self):
        return float(self._next())

    def readBoolean(self):
        return self.readInteger() != 0

    def readValueDictionaryId(self):
        return self.readInteger()

    def skip(self, n=1):
        for _ in range(n):
            self._next()

    def values(self):
        """All decoded values (not part of the reference interface; bulk access for columnar callers)."""
        return self._values


class RunLengthBitPackingHybridValuesReader(ValuesReader):
    """RunLengthBitPackingHybridValuesReader.java:31-66: 4-byte length prefix, then the hybrid stream."""
    _encoding = ENC_RLE
    _flags = F_LENGTH_PREFIX

    def __init__(self, bitWidth):
        super().__init__()
        self._bit_width = bitWidth

    def initFromPage(self, valueCount, stream):
        if stream.available() < 4:
            raise EOFException()
        super().initFromPage(valueCount, stream)

    def _consume(self, stream, r, start):
        if r.status in (EOF, DECODE) and r.values_written == 0 and r.bytes_consumed == 0:
            _raise_for(r.status)  # sliceStream(length) failed inside initFromPage
        stream.skipFully(r.bytes_consumed)
        self._next_offset = r.bytes_consumed


class RunLengthBitPackingHybridDecoder(ValuesReader):
    """Bare hybrid stream of known width (V2 levels: ColumnReaderBase.java:770-780)."""
    _encoding = ENC_RLE

    def __init__(self, bitWidth, stream=None, valueCount=0):
        super().__init__()
        if not 0 <= bitWidth <= 32:
            raise IllegalArgumentException("bitWidth must be >= 0 and <= 32")
        self._bit_width = bitWidth
        if stream is not None:
            self.initFromPage(valueCount, stream)

    readInt = ValuesReader.readInteger


class ZeroIntegerValuesReader(ValuesReader):
    """ZeroIntegerValuesReader.java:30-47: max level 0, consumes nothing, yields zeros."""

    def initFromPage(self, valueCount, stream):
        self._next_offset = 0

    def readInteger(self):
        return 0

    def skip(self, n=1):
        pass


class PlainBinaryDictionary:
    """PlainValuesDictionary.PlainBinaryDictionary (PlainValuesDictionary.java:59-135): the entries of
    a PLAIN BYTE_ARRAY dictionary page ([length LE32][bytes]...), or ``length`` bytes each when
    ``length`` is given (FIXED_LEN_BYTE_ARRAY; INT96 = 12, Encoding.java:97-103).  The page is parsed by
    ``pqd_binary_dict_views`` and kept on the device by a ``pqd_dict`` handle."""

    def __init__(self, dictionaryPageBytes, dictionarySize, length=None):
        if length is not None and length <= 0:
            raise IllegalArgumentException("Invalid byte array length: %s" % length)
        self._page = _np.ascontiguousarray(_np.frombuffer(bytes(dictionaryPageBytes), _np.uint8))
        self._n = dictionarySize
        self._views = _np.zeros(max(dictionarySize, 1), _np.uint64)
        st = lib.pqd_binary_dict_views(_ptr(self._page), self._page.size, dictionarySize, length or 0,
                                       self._views.ctypes.data_as(_C.c_void_p))
        if st == DECODE:
            raise IndexError("dictionary entry length outside the page")  # IndexOutOfBoundsException
        _raise_for(st, "PlainBinaryDictionary")
        self._views = self._views[:dictionarySize]
        kind = TYPE_BYTE_ARRAY if length is None else TYPE_FIXED_LEN_BYTE_ARRAY
        self._handle = _C.c_void_p()
        check(lib.pqd_dict_create(kind, length or 0, _ptr(self._page), self._page.size, dictionarySize,
                                  _C.byref(self._handle)), "pqd_dict_create")

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            lib.pqd_dict_destroy(h)

    def getMaxId(self):
        return self._n - 1

    def decodeToBinary(self, id):
        if not 0 <= id < self._n:
            raise IndexError(id)  # ArrayIndexOutOfBoundsException
        v = int(self._views[id])
        if v == VIEW_INVALID:
            raise IndexError("dictionary entry outside the page")
        off, ln = v & 0xFFFFFFFF, v >> 32
        return self._page[off:off + ln].tobytes()

    def max_length(self):
        ok = self._views[self._views != _np.uint64(VIEW_INVALID)]
        return int((ok >> _np.uint64(32)).max()) if ok.size else 0


class DictionaryValuesReader(ValuesReader):
    """DictionaryValuesReader.java:35-128; ``dictionary`` is the decoded PLAIN dictionary page: a
    numpy array of int32/int64/float32/float64 (PlainValuesDictionary.java:148-157,232-241), or a
    ``PlainBinaryDictionary`` (then ``readBytes`` is the accessor, :76-83)."""
    _encoding = ENC_RLE_DICTIONARY

    def __init__(self, dictionary):
        self._binary = dictionary if isinstance(dictionary, PlainBinaryDictionary) else None
        self._dict = None if self._binary else _np.ascontiguousarray(dictionary)
        self._dtype = _np.uint64 if self._binary else self._dict.dtype
        super().__init__()

    def _dict_array(self):
        return self._dict

    def initFromPage(self, valueCount, stream):
        if self._binary is None:
            return super().initFromPage(valueCount, stream)
        data = _np.ascontiguousarray(stream._peek_all())
        cap = max(valueCount * self._binary.max_length(), 1)
        self._offsets = _np.zeros(valueCount + 1, _np.int32)
        self._bytes = _np.zeros(cap, _np.uint8)
        r = PageResult()
        check(lib.pqd_decode_binary_page(self._binary._handle, _ptr(data), data.size, valueCount,
                                         _ptr(self._offsets), _ptr(self._bytes), cap, _C.byref(r)),
              "pqd_decode_binary_page")
        self._status = r.status
        self._values = _np.arange(r.values_written)  # positions; the bytes are cut out in readBytes
        self._i = 0
        stream.skipFully(stream.available())

    def readBytes(self):
        i = int(self._next())
        return self._bytes[self._offsets[i]:self._offsets[i + 1]].tobytes()

    def offsets_and_bytes(self):
        """Bulk access (not in the reference interface): offsets[n+1] and the value bytes."""
        n = len(self._values)
        return self._offsets[:n + 1], self._bytes[:self._offsets[n]]

    def _fail_at_end(self):
        if self._status == OK and len(self._values) == 0:
            raise ParquetDecodingException("Attempt to read from empty page")
        super()._fail_at_end()


class DictionaryIdReader(ValuesReader):
    """readValueDictionaryId() view of a dictionary-encoded page (DictionaryValuesReader.java:66-73)."""
    _encoding = ENC_RLE
    _flags = F_WIDTH_PREFIX


class DeltaBinaryPackingValuesReader(ValuesReader):
    """DeltaBinaryPackingValuesReader.java:33-173 (eager whole-page decode)."""
    _encoding = ENC_DELTA_BINARY_PACKED

    def __init__(self, dtype=_np.int64):
        self._dtype = _np.dtype(dtype).type
        super().__init__()

    def initFromPage(self, valueCount, stream):
        super().initFromPage(valueCount, stream)
        if self._status != OK:
            _raise_for(self._status)

    def _consume(self, stream, r, start):
        if r.status == OK:
            stream.skipFully(r.bytes_consumed)
            self._next_offset = r.bytes_consumed

    def _fail_at_end(self):
        raise ParquetDecodingException("no more value to read, total value count is %d" % len(self._values))

    def skip(self, n=1):
        if self._i + n > len(self._values):
            self._fail_at_end()
        self._i += n


class DeltaLengthByteArrayValuesReader(ValuesReader):
    """DeltaLengthByteArrayValuesReader.java:33-77: ``readBytes`` returns the next value; reading past the
    values the page holds raises ParquetDecodingException ("Failed to read N bytes" / "no more value to
    read")."""
    _encoding = ENC_DELTA_LENGTH_BYTE_ARRAY

    def _previous_value(self):
        return b""

    def initFromPage(self, valueCount, stream):
        data = _np.ascontiguousarray(stream._peek_all())
        prev = _np.frombuffer(self._previous_value(), _np.uint8)
        self._offsets = _np.zeros(valueCount + 1, _np.int32)
        r, need = PageResult(), _C.c_uint64(0)
        cap = max(4 * data.size, 1 << 12)
        while True:
            self._bytes = _np.zeros(cap, _np.uint8)
            check(lib.pqd_decode_delta_byte_array_page(self._encoding, _ptr(data), data.size, valueCount,
                                                       _ptr(prev) if prev.size else None, prev.size, _ptr(self._offsets),
                                                       _ptr(self._bytes), cap, _C.byref(r), _C.byref(need)),
                  "pqd_decode_delta_byte_array_page")
            if r.status != BAD_ARG or need.value <= cap:
                break
            cap = need.value  # the lengths asked for more room than the first guess
        if r.status != OK and r.values_written == 0 and r.bytes_consumed == 0 and self._encoding != ENC_PLAIN_BYTE_ARRAY:
            _raise_for(r.status, type(self).__name__ + ".initFromPage")  # a length block failed
        self._status = r.status
        self._values = _np.arange(r.values_written)
        self._i = 0
        stream.skipFully(r.bytes_consumed)

    def readBytes(self):
        i = int(self._next())
        return self._bytes[self._offsets[i]:self._offsets[i + 1]].tobytes()

    def offsets_and_bytes(self):
        """Bulk access (not in the reference interface): offsets[n+1] and the value bytes."""
        n = len(self._values)
        return self._offsets[:n + 1], self._bytes[:self._offsets[n]]

    def _fail_at_end(self):
        raise ParquetDecodingException("Failed to read the next value: the page holds %d" % len(self._values))

    def skip(self, n=1):
        for _ in range(n):
            self._next()


class BinaryPlainValuesReader(DeltaLengthByteArrayValuesReader):
    """BinaryPlainValuesReader.java:36-74: every value is its length (four bytes, little-endian) and its bytes;
    ``readBytes`` of a value that does not lie inside the page raises ParquetDecodingException ("could not read
    bytes at offset").  The whole page is decoded in ``initFromPage``; a failure shows at the value it belongs to."""
    _encoding = ENC_PLAIN_BYTE_ARRAY

    def _fail_at_end(self):
        raise ParquetDecodingException("could not read bytes at offset %d" % self._offsets[len(self._values)])


class DeltaByteArrayReader(DeltaLengthByteArrayValuesReader):
    """DeltaByteArrayReader.java:32-95 (prefix lengths, suffixes, the previous value carried from page to
    page through ``setPreviousReader``)."""
    _encoding = ENC_DELTA_BYTE_ARRAY

    def __init__(self):
        super().__init__()
        self._previous = b""

    def setPreviousReader(self, reader):
        """:87-95: the last value the previous page's reader delivered."""
        if reader is not None and len(getattr(reader, "_values", ())):
            n = len(reader._values)
            self._previous = reader._bytes[reader._offsets[n - 1]:reader._offsets[n]].tobytes()

    def _previous_value(self):
        return self._previous


class ByteStreamSplitValuesReader(ValuesReader):
    """ByteStreamSplitValuesReader.java:30-235 and its typed subclasses."""
    _encoding = ENC_BYTE_STREAM_SPLIT

    def __init__(self, dtype=None, length=None):
        if dtype is None:
            self._k = int(length)
            self._dtype = _np.uint8
        else:
            self._dtype = _np.dtype(dtype).type
            self._k = _np.dtype(dtype).itemsize
        super().__init__()

    def _elem_size(self):
        return self._k

    def initFromPage(self, valueCount, stream):
        super().initFromPage(valueCount, stream)
        if self._status != OK:
            _raise_for(self._status, "Invalid ByteStreamSplit stream")

    def _consume(self, stream, r, start):
        if r.status == OK:
            stream.skipFully(r.bytes_consumed)

    def _fail_at_end(self):
        raise ParquetDecodingException("Byte-stream data was already exhausted.")

    def readBytes(self):
        return bytes(self._next())

    def skip(self, n=1):
        if n < 0 or self._i + n > len(self._values):
            raise ParquetDecodingException(
                "Cannot skip this many elements. Current index: %d. Skip %d. Total number of elements: %d" %
                (self._i, n, len(self._values)))
        self._i += n


class ByteBitPackingValuesReader(ValuesReader):
    """ByteBitPackingValuesReader.java:29-94 with Packer.BIG_ENDIAN (legacy BIT_PACKED levels)."""
    _encoding = ENC_BIT_PACKED

    def __init__(self, bound):
        super().__init__()
        self._bit_width = int(bound).bit_length()  # BytesUtils.getWidthFromMaxInt

    def _consume(self, stream, r, start):
        stream.skipFully(r.bytes_consumed)
        self._next_offset = r.bytes_consumed

    def _fail_at_end(self):
        raise IndexError("read past the page's value count")


# ------------------------------------------------------------------------------- device-resident batches
class PlainValuesReader(ValuesReader):
    """PlainValuesReader.{Integer,Long,Float,Double}PlainValuesReader (parquet-column/.../values/plain/
    PlainValuesReader.java:38-145): initFromPage takes everything that is available, each read returns the
    next little-endian value, a read past the end raises ParquetDecodingException("could not read ...")."""
    _encoding = ENC_PLAIN

    def __init__(self, dtype):
        self._dtype = _np.dtype(dtype)
        super().__init__()

    def _fail_at_end(self):
        raise ParquetDecodingException("could not read " + {"int32": "int", "int64": "long", "float32": "float",
                                                             "float64": "double"}.get(self._dtype.name, "value"))

    def skip(self, n=1):
        if self._i + n > len(self._values) or n < 0:
            raise ParquetDecodingException("could not skip %d values" % n)
        self._i += n

    class IntegerPlainValuesReader:
        def __new__(cls):
            return PlainValuesReader(_np.int32)

    class LongPlainValuesReader:
        def __new__(cls):
            return PlainValuesReader(_np.int64)

    class FloatPlainValuesReader:
        def __new__(cls):
            return PlainValuesReader(_np.float32)

    class DoublePlainValuesReader:
        def __new__(cls):
            return PlainValuesReader(_np.float64)


class BooleanPlainValuesReader(ValuesReader):
    """BooleanPlainValuesReader (parquet-column/.../values/plain/BooleanPlainValuesReader.java:38-89): one bit
    per value, LSB first; the page is min(ceil(valueCount / 8), available) bytes long."""
    _encoding = ENC_PLAIN_BOOLEAN
    _dtype = _np.uint8

    def _consume(self, stream, r, start):
        stream.skipFully(r.bytes_consumed)
        self._next_offset = start + r.bytes_consumed

    def _fail_at_end(self):
        raise ParquetDecodingException("attempt to read beyond end of boolean page")

    def readBoolean(self):
        return bool(self._next())

    def skip(self, n=1):
        self._i += n  # the reference only moves its bit index; a later read fails


class BytesDecompressor:
    """CodecFactory.BytesDecompressor (parquet-hadoop/.../hadoop/CodecFactory.java:165-217) for the codecs
    the device handles: ``decompress(bytes, decompressedSize)`` returns exactly ``decompressedSize``
    bytes or raises IOError, which ColumnChunkPageReadStore turns into "could not decompress page"."""
    _codecs = {"UNCOMPRESSED": CODEC_UNCOMPRESSED, "SNAPPY": CODEC_SNAPPY, "LZ4_RAW": CODEC_LZ4_RAW, "ZSTD": CODEC_ZSTD, "GZIP": CODEC_GZIP}

    def __init__(self, codecName):
        if codecName not in self._codecs or not lib.pqd_codec_supported(self._codecs[codecName]):
            raise IllegalArgumentException("no device decompressor for " + str(codecName))
        self._codec = self._codecs[codecName]

    def decompress(self, data, decompressedSize):
        src = _np.frombuffer(bytes(data), _np.uint8)
        dst = _np.zeros(max(decompressedSize, 1), _np.uint8)
        r = PageResult()
        check(lib.pqd_decompress_page(self._codec, _ptr(src) if src.size else None, src.size, _ptr(dst), decompressedSize,
                                      _C.byref(r)), "pqd_decompress_page")
        if r.status != OK:
            raise IOError("could not decompress page: %d of %d bytes" % (r.values_written, decompressedSize))
        return dst[:decompressedSize].tobytes()

    def release(self):
        pass


def make_descs(pages):
    """Builds a host array of ``PageDesc`` from dicts with keys data/out/dict (device addresses as
    ints), data_len, num_values, dict_len, encoding, elem_size, bit_width, flags."""
    arr = (PageDesc * len(pages))()
    for i, p in enumerate(pages):
        d = arr[i]
        d.data = p.get("data") or None
        d.out = p.get("out") or None
        d.dict = p.get("dict") or None
        d.data_len = p.get("data_len", 0)
        d.num_values = p.get("num_values", 0)
        d.dict_len = p.get("dict_len", 0)
        d.encoding = p["encoding"]
        d.elem_size = p.get("elem_size", 4)
        d.bit_width = p.get("bit_width", 0)
        d.flags = p.get("flags", 0)
    return arr


def descs_to_numpy(arr):
    return _np.frombuffer(bytes(arr), dtype=_np.uint8).copy()


def results_from_numpy(raw):
    r = _np.ascontiguousarray(raw).view(_np.uint32).reshape(-1, 4)
    return r[:, 0].astype(_np.int32), r[:, 1].copy(), r[:, 2].copy()


def decode_pages_dev(descs_dev_ptr, results_dev_ptr, n_pages, encoding, elem_size, stream=None):
    """pqd_decode_pages_dev: descriptors and results already on the device; asynchronous on ``stream``."""
    check(lib.pqd_decode_pages_dev(descs_dev_ptr, results_dev_ptr, n_pages, encoding, elem_size, stream),
          "pqd_decode_pages_dev")
