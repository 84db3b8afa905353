// C++ host mirror of the parquet-java reader surface for the page-decode path, on top of the C ABI
// in pqd.h (header-only; link with libpqdecode_hip.so).  Same class and method names, argument
// meaning and failure behaviour as the Java classes cited at each declaration, so that code (and
// tests) written against parquet-java's readers reads the same here.  Every read is served from
// values decoded on the device by ONE pqd_decode_page call made in initFromPage.
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "pqd.h"

namespace parquet_hip {

// java.io.EOFException / IllegalArgumentException / org.apache.parquet.io.ParquetDecodingException
struct EOFException : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct IllegalArgumentException : std::invalid_argument {
  using std::invalid_argument::invalid_argument;
};
struct ParquetDecodingException : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// Rethrows a library status as the exception the Java decoder throws for it.
inline void raise_for(int status, const std::string& what = "") {
  switch (status) {
    case PQD_OK: return;
    case PQD_EOF: throw EOFException(what.empty() ? "EOFException" : what);
    case PQD_BAD_ARG: throw IllegalArgumentException(what + ": " + pqd_last_error());
    case PQD_PAST_END: throw IllegalArgumentException("Reading past RLE/BitPacking stream.");
    case PQD_DECODE:
    case PQD_DICT_OOB: throw ParquetDecodingException(what + ": " + pqd_status_string(status));
    default: throw std::runtime_error(what + ": " + pqd_status_string(status) + " [" + pqd_last_error() + "]");
  }
}

// org.apache.parquet.bytes.ByteBufferInputStream (parquet-common/.../bytes/ByteBufferInputStream.java):
// a cursor over bytes the caller owns.
class ByteBufferInputStream {
 public:
  ByteBufferInputStream(const uint8_t* data, size_t len) : data_(data), len_(len) {}
  static ByteBufferInputStream wrap(const std::vector<uint8_t>& v) { return ByteBufferInputStream(v.data(), v.size()); }
  size_t position() const { return pos_; }
  size_t available() const { return len_ - pos_; }
  const uint8_t* current() const { return data_ + pos_; }
  void skipFully(size_t n) {
    if (n > available()) throw EOFException("skipFully past the end");
    pos_ += n;
  }
  // slice(length): the next `length` bytes, consumed; EOFException if fewer remain
  const uint8_t* slice(size_t length) {
    if (length > available()) throw EOFException("slice past the end");
    const uint8_t* p = data_ + pos_;
    pos_ += length;
    return p;
  }

 private:
  const uint8_t* data_;
  size_t len_, pos_ = 0;
};

// ParquetReadRouter (parquet-plugins/parquet-encoding-vector/.../ParquetReadRouter.java:57-116): bulk
// unpack of one bit-packed run; consumes exactly currentCount * bitWidth / 8 bytes.
struct ParquetReadRouter {
  static void read(int bitWidth, ByteBufferInputStream& in, int currentCount, int32_t* currentBuffer) {
    const size_t bytes = (size_t)currentCount / 8 * (size_t)bitWidth;
    const uint8_t* src = in.slice(bytes);
    raise_for(pqd_unpack_i32(bitWidth, src, bytes, currentBuffer, (size_t)currentCount), "ParquetReadRouter.read");
  }
  static void readBatchUsing512Vector(int bitWidth, ByteBufferInputStream& in, int currentCount, int32_t* currentBuffer) {
    read(bitWidth, in, currentCount, currentBuffer);
  }
  static void readBatch(int bitWidth, ByteBufferInputStream& in, int currentCount, int32_t* currentBuffer) {
    read(bitWidth, in, currentCount, currentBuffer);
  }
  // BytePackerForLong.unpack8Values over a whole run (parquet-encoding/.../BytePackerForLong.java:72-105)
  static void readLongs(int bitWidth, ByteBufferInputStream& in, int currentCount, int64_t* currentBuffer) {
    const size_t bytes = (size_t)currentCount / 8 * (size_t)bitWidth;
    const uint8_t* src = in.slice(bytes);
    raise_for(pqd_unpack_i64(bitWidth, src, bytes, currentBuffer, (size_t)currentCount), "ParquetReadRouter.readLongs");
  }
};

// BytePacker (parquet-encoding/.../bitpacking/BytePacker.java:40-148) as handed out by
// Packer.LITTLE_ENDIAN.newBytePacker(width) / newBytePackerVector(width) (Packer.java:92-121): the
// per-step interface.  Every call is one device call; bulk callers use ParquetReadRouter.
class BytePacker {
 public:
  explicit BytePacker(int bitWidth) : w_(bitWidth) {
    if (bitWidth < 0 || bitWidth > 32) throw IllegalArgumentException("bitWidth must be in 0..32");
  }
  int getBitWidth() const { return w_; }
  // ByteBitPacking512VectorLE: 32 values per step for widths 1-7 and 9-15, 64 for 8, 16 for 16-32
  int getUnpackCount() const { return w_ == 0 ? 0 : w_ == 8 ? 64 : w_ < 16 ? 32 : 16; }
  void unpack8Values(const uint8_t* input, int inPos, int32_t* output, int outPos) const { unpack(8, input, inPos, output, outPos); }
  void unpack32Values(const uint8_t* input, int inPos, int32_t* output, int outPos) const { unpack(32, input, inPos, output, outPos); }
  void unpackValuesUsingVector(const uint8_t* input, int inPos, int32_t* output, int outPos) const {
    unpack(getUnpackCount(), input, inPos, output, outPos);
  }

 private:
  void unpack(int n, const uint8_t* input, int inPos, int32_t* output, int outPos) const {
    if (n == 0) return;
    raise_for(pqd_unpack_i32(w_, input + inPos, (size_t)n * (size_t)w_ / 8, output + outPos, (size_t)n), "BytePacker.unpack");
  }
  int w_;
};

// BytePackerForLong (parquet-encoding/.../bitpacking/BytePackerForLong.java:72-105), widths 0..64.
class BytePackerForLong {
 public:
  explicit BytePackerForLong(int bitWidth) : w_(bitWidth) {
    if (bitWidth < 0 || bitWidth > 64) throw IllegalArgumentException("bitWidth must be in 0..64");
  }
  int getBitWidth() const { return w_; }
  void unpack8Values(const uint8_t* input, int inPos, int64_t* output, int outPos) const { unpack(8, input, inPos, output, outPos); }
  void unpack32Values(const uint8_t* input, int inPos, int64_t* output, int outPos) const { unpack(32, input, inPos, output, outPos); }

 private:
  void unpack(int n, const uint8_t* input, int inPos, int64_t* output, int outPos) const {
    raise_for(pqd_unpack_i64(w_, input + inPos, (size_t)n * (size_t)w_ / 8, output + outPos, (size_t)n), "BytePackerForLong.unpack");
  }
  int w_;
};

// ValuesReader (parquet-column/.../values/ValuesReader.java:107-202).
class ValuesReader {
 public:
  virtual ~ValuesReader() = default;

  virtual void initFromPage(int valueCount, ByteBufferInputStream& stream) {
    decoded_.assign((size_t)(valueCount > 0 ? valueCount : 1) * elem_size_, 0);
    pqd_page_desc d;
    std::memset(&d, 0, sizeof d);
    d.data = stream.available() ? stream.current() : nullptr;
    d.data_len = (uint32_t)stream.available();
    d.out = decoded_.data();
    d.num_values = (uint32_t)valueCount;
    d.dict = dict_;
    d.dict_len = dict_len_;
    d.encoding = (uint8_t)encoding_;
    d.elem_size = (uint8_t)elem_size_;
    d.bit_width = (uint8_t)bit_width_;
    d.flags = (uint8_t)flags_;
    pqd_page_result r;
    std::memset(&r, 0, sizeof r);
    raise_for(pqd_decode_page(&d, &r), "pqd_decode_page");
    status_ = r.status;
    count_ = r.values_written;
    next_ = 0;
    consume(stream, r);
  }

  int readInteger() { return (int)load<int32_t>(); }
  int64_t readLong() { return load<int64_t>(); }
  float readFloat() { return load<float>(); }
  double readDouble() { return load<double>(); }
  bool readBoolean() { return readInteger() != 0; }
  int readValueDictionaryId() { return readInteger(); }
  virtual void skip() { index(); }
  virtual void skip(int n) {
    for (int i = 0; i < n; ++i) index();
  }
  // getNextOffset(): where the next section of the page starts, for the readers that define it
  size_t getNextOffset() const {
    if (!has_next_offset_) throw ParquetDecodingException("Unsupported: cannot get offset of the next section.");
    return next_offset_;
  }
  // all decoded values (not part of the reference interface; bulk access for columnar callers)
  const uint8_t* values() const { return decoded_.data(); }
  size_t valuesWritten() const { return count_; }
  int pageStatus() const { return status_; }

 protected:
  ValuesReader(int encoding, int elem_size) : encoding_(encoding), elem_size_(elem_size) {}
  virtual void consume(ByteBufferInputStream& stream, const pqd_page_result&) { stream.skipFully(stream.available()); }
  virtual void failAtEnd() { raise_for(status_ != PQD_OK ? status_ : PQD_PAST_END); }
  size_t index() {
    if (next_ >= count_) failAtEnd();
    return next_++;
  }
  template <typename T>
  T load() {
    if ((int)sizeof(T) != elem_size_) throw std::logic_error("read method does not match the page's value type");
    T v;
    std::memcpy(&v, decoded_.data() + index() * sizeof(T), sizeof(T));
    return v;
  }

  int encoding_, elem_size_, bit_width_ = 0, flags_ = 0;
  const void* dict_ = nullptr;
  uint32_t dict_len_ = 0;
  std::vector<uint8_t> decoded_;
  size_t count_ = 0, next_ = 0, next_offset_ = 0;
  bool has_next_offset_ = false;
  int status_ = PQD_OK;
};

// RunLengthBitPackingHybridValuesReader (.../values/rle/RunLengthBitPackingHybridValuesReader.java:31-66):
// 4-byte little-endian length, then the hybrid stream.
class RunLengthBitPackingHybridValuesReader : public ValuesReader {
 public:
  explicit RunLengthBitPackingHybridValuesReader(int bitWidth) : ValuesReader(PQD_ENC_RLE, 4) {
    bit_width_ = bitWidth;
    flags_ = PQD_F_LENGTH_PREFIX;
  }
  void initFromPage(int valueCount, ByteBufferInputStream& stream) override {
    if (stream.available() < 4) throw EOFException("length prefix");
    ValuesReader::initFromPage(valueCount, stream);
  }

 protected:
  void consume(ByteBufferInputStream& stream, const pqd_page_result& r) override {
    if ((r.status == PQD_EOF || r.status == PQD_DECODE) && r.values_written == 0 && r.bytes_consumed == 0)
      raise_for(r.status, "sliceStream(length)");
    stream.skipFully(r.bytes_consumed);
    next_offset_ = r.bytes_consumed;
    has_next_offset_ = true;
  }
};

// RunLengthBitPackingHybridDecoder (.../values/rle/RunLengthBitPackingHybridDecoder.java:52-109): a bare
// hybrid stream of known width (V2 levels).
class RunLengthBitPackingHybridDecoder : public ValuesReader {
 public:
  explicit RunLengthBitPackingHybridDecoder(int bitWidth) : ValuesReader(PQD_ENC_RLE, 4) {
    if (bitWidth < 0 || bitWidth > 32) throw IllegalArgumentException("bitWidth must be >= 0 and <= 32");
    bit_width_ = bitWidth;
  }
  int readInt() { return readInteger(); }
};

// DictionaryValuesReader (.../values/dictionary/DictionaryValuesReader.java:44-127) over a fixed-width
// dictionary: `dictionary` = the PLAIN dictionary page body (PlainValuesDictionary.java:148-157,232-241).
class DictionaryValuesReader : public ValuesReader {
 public:
  DictionaryValuesReader(const void* dictionary, uint32_t dictionarySize, int elemSize)
      : ValuesReader(PQD_ENC_RLE_DICTIONARY, elemSize) {
    dict_ = dictionary;
    dict_len_ = dictionarySize;
  }

 protected:
  void failAtEnd() override {
    if (status_ == PQD_OK && count_ == 0) throw ParquetDecodingException("Attempt to read from empty page");
    ValuesReader::failAtEnd();
  }
};

// PlainValuesDictionary.PlainBinaryDictionary (.../values/dictionary/PlainValuesDictionary.java:59-135):
// a PLAIN dictionary page of BYTE_ARRAY entries, or of `length`-byte entries (FIXED_LEN_BYTE_ARRAY;
// INT96 = 12, Encoding.java:97-103).  The page is walked by pqd_binary_dict_views and kept on the
// device by a pqd_dict handle.
class PlainBinaryDictionary {
 public:
  PlainBinaryDictionary(const uint8_t* dictionaryPage, size_t pageLength, uint32_t dictionarySize, int length = 0)
      : page_(dictionaryPage, dictionaryPage + pageLength), views_(dictionarySize ? dictionarySize : 1) {
    if (length < 0) throw IllegalArgumentException("Invalid byte array length");
    int st = pqd_binary_dict_views(page_.data(), page_.size(), dictionarySize, length, views_.data());
    if (st == PQD_DECODE) throw std::out_of_range("dictionary entry length outside the page");  // IndexOutOfBoundsException
    raise_for(st, "PlainBinaryDictionary");
    views_.resize(dictionarySize);
    raise_for(pqd_dict_create(length ? PQD_TYPE_FIXED_LEN_BYTE_ARRAY : PQD_TYPE_BYTE_ARRAY, length, page_.data(),
                              page_.size(), dictionarySize, &handle_),
              "pqd_dict_create");
  }
  PlainBinaryDictionary(const PlainBinaryDictionary&) = delete;
  PlainBinaryDictionary& operator=(const PlainBinaryDictionary&) = delete;
  ~PlainBinaryDictionary() { pqd_dict_destroy(handle_); }

  int getMaxId() const { return (int)views_.size() - 1; }
  std::string decodeToBinary(int id) const {
    const uint64_t v = views_.at((size_t)id);  // std::out_of_range = ArrayIndexOutOfBoundsException
    if (v == PQD_VIEW_INVALID) throw std::out_of_range("dictionary entry outside the page");
    return std::string((const char*)page_.data() + (uint32_t)v, (size_t)(v >> 32));
  }
  uint32_t maxLength() const {
    uint32_t m = 0;
    for (uint64_t v : views_)
      if (v != PQD_VIEW_INVALID && (uint32_t)(v >> 32) > m) m = (uint32_t)(v >> 32);
    return m;
  }
  const pqd_dict* handle() const { return handle_; }

 private:
  std::vector<uint8_t> page_;
  std::vector<uint64_t> views_;
  pqd_dict* handle_ = nullptr;
};

// DictionaryValuesReader over a PlainBinaryDictionary: readBytes() (DictionaryValuesReader.java:76-83).
class BinaryDictionaryValuesReader {
 public:
  explicit BinaryDictionaryValuesReader(const PlainBinaryDictionary& dictionary) : dict_(dictionary) {}
  void initFromPage(int valueCount, ByteBufferInputStream& stream) {
    const uint32_t n = (uint32_t)(valueCount > 0 ? valueCount : 0);
    const uint64_t cap64 = (uint64_t)n * dict_.maxLength();
    const uint32_t cap = cap64 > 0x7fffffffu ? 0x7fffffffu : (uint32_t)cap64;
    offsets_.assign((size_t)n + 1, 0);
    bytes_.assign(cap ? cap : 1, 0);
    pqd_page_result r;
    std::memset(&r, 0, sizeof r);
    raise_for(pqd_decode_binary_page(dict_.handle(), stream.available() ? stream.current() : nullptr,
                                     (uint32_t)stream.available(), n, offsets_.data(), bytes_.data(), cap, &r),
              "pqd_decode_binary_page");
    status_ = r.status;
    count_ = r.values_written;
    next_ = 0;
    stream.skipFully(stream.available());
  }
  std::string readBytes() {
    const size_t i = index();
    return std::string((const char*)bytes_.data() + offsets_[i], (size_t)(offsets_[i + 1] - offsets_[i]));
  }
  void skip() { index(); }
  void skip(int n) {
    for (int i = 0; i < n; ++i) index();
  }
  // bulk access (not part of the reference interface): offsets[valuesWritten()+1] and the value bytes
  const int32_t* offsets() const { return offsets_.data(); }
  const uint8_t* bytes() const { return bytes_.data(); }
  size_t valuesWritten() const { return count_; }

 private:
  size_t index() {
    if (next_ >= count_) {
      if (status_ == PQD_OK && count_ == 0) throw ParquetDecodingException("Attempt to read from empty page");
      raise_for(status_ != PQD_OK ? status_ : PQD_PAST_END);
    }
    return next_++;
  }
  const PlainBinaryDictionary& dict_;
  std::vector<int32_t> offsets_;
  std::vector<uint8_t> bytes_;
  size_t count_ = 0, next_ = 0;
  int status_ = PQD_OK;
};

// readValueDictionaryId() view of a dictionary-encoded page (DictionaryValuesReader.java:66-73).
class DictionaryIdReader : public ValuesReader {
 public:
  DictionaryIdReader() : ValuesReader(PQD_ENC_RLE, 4) { flags_ = PQD_F_WIDTH_PREFIX; }
};

// DeltaBinaryPackingValuesReader (.../values/delta/DeltaBinaryPackingValuesReader.java:60-172).
class DeltaBinaryPackingValuesReader : public ValuesReader {
 public:
  explicit DeltaBinaryPackingValuesReader(int elemSize = 8) : ValuesReader(PQD_ENC_DELTA_BINARY_PACKED, elemSize) {}
  void initFromPage(int valueCount, ByteBufferInputStream& stream) override {
    ValuesReader::initFromPage(valueCount, stream);
    if (status_ != PQD_OK) raise_for(status_, "DeltaBinaryPackingValuesReader");
  }
  void skip(int n) override {
    if (next_ + (size_t)n > count_) failAtEnd();
    next_ += (size_t)n;
  }
  void skip() override { skip(1); }

 protected:
  void consume(ByteBufferInputStream& stream, const pqd_page_result& r) override {
    if (r.status != PQD_OK) return;
    stream.skipFully(r.bytes_consumed);
    next_offset_ = r.bytes_consumed;
    has_next_offset_ = true;
  }
  void failAtEnd() override {
    throw ParquetDecodingException("no more value to read, total value count is " + std::to_string(count_));
  }
};

// ByteStreamSplitValuesReader and its typed subclasses (.../bytestreamsplit/ByteStreamSplitValuesReader.java:53-218).
class ByteStreamSplitValuesReader : public ValuesReader {
 public:
  explicit ByteStreamSplitValuesReader(int elementSizeInBytes) : ValuesReader(PQD_ENC_BYTE_STREAM_SPLIT, elementSizeInBytes) {}
  void initFromPage(int valueCount, ByteBufferInputStream& stream) override {
    ValuesReader::initFromPage(valueCount, stream);
    if (status_ != PQD_OK) raise_for(status_, "Invalid ByteStreamSplit stream");
  }
  // readBytes() of the FIXED_LEN_BYTE_ARRAY reader: a view into the decoded buffer
  const uint8_t* readBytes() { return decoded_.data() + index() * (size_t)elem_size_; }
  void skip(int n) override {
    if (n < 0 || next_ + (size_t)n > count_)
      throw ParquetDecodingException("Cannot skip this many elements. Current index: " + std::to_string(next_) + ". Skip " +
                                     std::to_string(n) + ". Total number of elements: " + std::to_string(count_));
    next_ += (size_t)n;
  }
  void skip() override { skip(1); }

 protected:
  void consume(ByteBufferInputStream& stream, const pqd_page_result& r) override {
    if (r.status == PQD_OK) stream.skipFully(r.bytes_consumed);
  }
  void failAtEnd() override { throw ParquetDecodingException("Byte-stream data was already exhausted."); }
};

// ByteBitPackingValuesReader with Packer.BIG_ENDIAN: legacy BIT_PACKED levels
// (.../values/bitpacking/ByteBitPackingValuesReader.java:41-88); width = bits of `bound`.
class ByteBitPackingValuesReader : public ValuesReader {
 public:
  explicit ByteBitPackingValuesReader(int bound) : ValuesReader(PQD_ENC_BIT_PACKED, 4) {
    int w = 0;
    for (unsigned b = (unsigned)bound; b; b >>= 1) ++w;  // BytesUtils.getWidthFromMaxInt
    bit_width_ = w;
  }

 protected:
  void consume(ByteBufferInputStream& stream, const pqd_page_result& r) override {
    stream.skipFully(r.bytes_consumed);
    next_offset_ = r.bytes_consumed;
    has_next_offset_ = true;
  }
  void failAtEnd() override { throw std::out_of_range("read past the page's value count"); }
};

}  // namespace parquet_hip
