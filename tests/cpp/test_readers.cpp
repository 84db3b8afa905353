// Exercises the C++ reader mirror (include/pqd_readers.hpp) against known answers taken from the
// reference's own tests.  Needs a gfx950 device; built by the Makefile, run by tests/test_cpp_mirror.py.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "pqd_readers.hpp"

using namespace parquet_hip;

static int failures = 0;
#define EXPECT(cond)                                                   \
  do {                                                                 \
    if (!(cond)) {                                                     \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);      \
      ++failures;                                                      \
    }                                                                  \
  } while (0)
template <typename E, typename F>
static bool throws(F f) {
  try {
    f();
  } catch (const E&) {
    return true;
  } catch (...) {
    return false;
  }
  return false;
}

// LSB-first packing straight from the definition (value i occupies bits [i*w, (i+1)*w))
static std::vector<uint8_t> pack_lsb(const std::vector<uint32_t>& v, int w) {
  std::vector<uint8_t> out((v.size() * w + 7) / 8, 0);
  for (size_t i = 0; i < v.size(); ++i)
    for (int b = 0; b < w; ++b)
      if ((v[i] >> b) & 1u) out[(i * w + b) / 8] |= (uint8_t)(1u << ((i * w + b) % 8));
  return out;
}

int main() {
  if (pqd_device_count() < 1) {
    std::printf("no gfx950 device\n");
    return 2;
  }
  EXPECT(pqd_init(0) == PQD_OK);

  // ParquetReadRouter: every width, 64 values; consumes exactly n*w/8 bytes; EOFException when short
  for (int w = 0; w <= 32; ++w) {
    std::vector<uint32_t> v(64);
    for (size_t i = 0; i < v.size(); ++i) v[i] = (uint32_t)((i * 2654435761u) & (w == 32 ? 0xFFFFFFFFu : ((1u << w) - 1u)));
    std::vector<uint8_t> packed = pack_lsb(v, w);
    packed.push_back(0xEE);
    ByteBufferInputStream in = ByteBufferInputStream::wrap(packed);
    std::vector<int32_t> out(64, -1);
    ParquetReadRouter::readBatchUsing512Vector(w, in, 64, out.data());
    EXPECT(in.position() == (size_t)(8 * w));
    for (size_t i = 0; i < v.size(); ++i) EXPECT((uint32_t)out[i] == v[i]);
  }
  {
    std::vector<uint8_t> few(5, 0);
    ByteBufferInputStream in = ByteBufferInputStream::wrap(few);
    std::vector<int32_t> out(8);
    EXPECT(throws<EOFException>([&] { ParquetReadRouter::read(7, in, 8, out.data()); }));
  }

  // TestRunLengthBitPackingHybridEncoder.testGroupBoundary: a truncated bit-packed group still decodes
  {
    std::vector<uint8_t> page = {(1 << 1) | 1, (1 << 0) | (2 << 2) | (3 << 4)};
    ByteBufferInputStream in = ByteBufferInputStream::wrap(page);
    RunLengthBitPackingHybridDecoder dec(2);
    dec.initFromPage(9, in);
    EXPECT(dec.readInt() == 1 && dec.readInt() == 2 && dec.readInt() == 3);
    for (int i = 0; i < 5; ++i) EXPECT(dec.readInt() == 0);
    EXPECT(throws<IllegalArgumentException>([&] { dec.readInt(); }));  // "Reading past RLE/BitPacking stream."
  }
  EXPECT(throws<IllegalArgumentException>([] { RunLengthBitPackingHybridDecoder bad(33); }));

  // V1 levels: 4-byte length, RLE run of 100 x 1, bit-packed group; getNextOffset
  {
    std::vector<uint8_t> body = {0xC8, 0x01, 1, (1 << 1) | 1, 0b10101010};  // varint(100 << 1) takes two bytes
    std::vector<uint8_t> page = {(uint8_t)body.size(), 0, 0, 0};
    page.insert(page.end(), body.begin(), body.end());
    page.push_back(0x77);  // next section
    ByteBufferInputStream in = ByteBufferInputStream::wrap(page);
    RunLengthBitPackingHybridValuesReader levels(1);
    levels.initFromPage(108, in);
    EXPECT(levels.getNextOffset() == 9 && in.position() == 9);
    for (int i = 0; i < 100; ++i) EXPECT(levels.readInteger() == 1);
    for (int i = 0; i < 8; ++i) EXPECT(levels.readBoolean() == ((i & 1) != 0));
  }

  // DictionaryValuesReader: TestDictionary-style i % 50 ids over a long dictionary, width prefix 6
  {
    std::vector<int64_t> dict(50);
    for (int i = 0; i < 50; ++i) dict[i] = (int64_t)i * 1000000007LL - 5;
    std::vector<uint32_t> ids(2000);
    for (size_t i = 0; i < ids.size(); ++i) ids[i] = (uint32_t)(i % 50);
    std::vector<uint8_t> page = {6};
    for (size_t g = 0; g < ids.size(); g += 504) {  // bit-packed runs of at most 63 groups, 1-byte headers
      size_t n = std::min<size_t>(504, ids.size() - g);
      size_t groups = (n + 7) / 8;
      page.push_back((uint8_t)((groups << 1) | 1));
      std::vector<uint32_t> part(ids.begin() + g, ids.begin() + g + n);
      part.resize(groups * 8, 0);
      std::vector<uint8_t> p = pack_lsb(part, 6);
      page.insert(page.end(), p.begin(), p.end());
    }
    ByteBufferInputStream in = ByteBufferInputStream::wrap(page);
    DictionaryValuesReader r(dict.data(), (uint32_t)dict.size(), 8);
    r.initFromPage((int)ids.size(), in);
    bool ok = true;
    for (size_t i = 0; i < ids.size(); ++i) ok = ok && r.readLong() == dict[i % 50];
    EXPECT(ok);
    // an id beyond the dictionary fails at that read, not before
    std::vector<uint8_t> bad = {6, (1 << 1) | 1};
    std::vector<uint32_t> bid = {0, 1, 60, 2, 0, 0, 0, 0};
    std::vector<uint8_t> bp = pack_lsb(bid, 6);
    bad.insert(bad.end(), bp.begin(), bp.end());
    ByteBufferInputStream in2 = ByteBufferInputStream::wrap(bad);
    DictionaryValuesReader r2(dict.data(), (uint32_t)dict.size(), 8);
    r2.initFromPage(8, in2);
    EXPECT(r2.readLong() == dict[0] && r2.readLong() == dict[1]);
    EXPECT(throws<ParquetDecodingException>([&] { r2.readLong(); }));
    // empty page
    std::vector<uint8_t> none;
    ByteBufferInputStream in3 = ByteBufferInputStream::wrap(none);
    DictionaryValuesReader r3(dict.data(), (uint32_t)dict.size(), 8);
    r3.initFromPage(0, in3);
    EXPECT(throws<ParquetDecodingException>([&] { r3.readLong(); }));
  }

  // BytePacker per-step interface: the vector step equals the scalar steps (TestByteBitPacking512VectorLE)
  {
    std::vector<uint8_t> raw(300);
    for (size_t i = 0; i < raw.size(); ++i) raw[i] = (uint8_t)(i * 37 + 11);
    bool ok = true;
    for (int w = 0; w <= 32; ++w) {
      BytePacker p(w);
      int cnt = p.getUnpackCount();
      std::vector<int32_t> a(64 + 2, -7), b(64, -7);
      p.unpackValuesUsingVector(raw.data(), 3, a.data(), 1);
      for (int g = 0; g < cnt / 8; ++g) p.unpack8Values(raw.data(), 3 + g * w, b.data(), g * 8);
      for (int i = 0; i < cnt; ++i) ok = ok && a[1 + i] == b[i];
      ok = ok && a[0] == -7 && a[1 + cnt] == -7;
      if (w > 0 && w < 32) ok = ok && (uint32_t)b[0] == ((raw[3] | raw[4] << 8 | raw[5] << 16 | (uint32_t)raw[6] << 24 | (uint64_t)raw[7] << 32) & ((1ull << w) - 1));
    }
    EXPECT(ok);
    BytePackerForLong pl(40);
    int64_t out[8];
    pl.unpack8Values(raw.data(), 0, out, 0);
    uint64_t v0 = 0;
    for (int b = 0; b < 5; ++b) v0 |= (uint64_t)raw[b] << (8 * b);
    EXPECT((uint64_t)out[0] == v0);
  }

  // PlainBinaryDictionary + readBytes: TestDictionary.testBinaryDictionary-style "a" + i % 50 strings
  {
    std::vector<std::string> words(50);
    std::vector<uint8_t> dpage;
    for (int i = 0; i < 50; ++i) {
      words[i] = "a" + std::to_string(i) + std::string((size_t)(i % 7), 'x');
      uint32_t l = (uint32_t)words[i].size();
      for (int b = 0; b < 4; ++b) dpage.push_back((uint8_t)(l >> (8 * b)));
      dpage.insert(dpage.end(), words[i].begin(), words[i].end());
    }
    PlainBinaryDictionary dict(dpage.data(), dpage.size(), 50);
    EXPECT(dict.getMaxId() == 49 && dict.decodeToBinary(13) == words[13]);
    EXPECT(throws<std::out_of_range>([&] { dict.decodeToBinary(50); }));
    std::vector<uint32_t> ids(3000);
    for (size_t i = 0; i < ids.size(); ++i) ids[i] = (uint32_t)((i * 7) % 50);
    std::vector<uint8_t> page = {6};
    for (size_t g = 0; g < ids.size(); g += 504) {
      size_t n = std::min<size_t>(504, ids.size() - g);
      size_t groups = (n + 7) / 8;
      page.push_back((uint8_t)((groups << 1) | 1));
      std::vector<uint32_t> part(ids.begin() + g, ids.begin() + g + n);
      part.resize(groups * 8, 0);
      std::vector<uint8_t> p = pack_lsb(part, 6);
      page.insert(page.end(), p.begin(), p.end());
    }
    ByteBufferInputStream in = ByteBufferInputStream::wrap(page);
    BinaryDictionaryValuesReader r(dict);
    r.initFromPage((int)ids.size(), in);
    bool ok = r.valuesWritten() == ids.size();
    for (size_t i = 0; i < ids.size(); ++i) ok = ok && r.readBytes() == words[ids[i]];
    EXPECT(ok);
    EXPECT(throws<IllegalArgumentException>([&] { r.readBytes(); }));  // "Reading past RLE/BitPacking stream."
    // fixed-length entries; a length prefix outside the page
    PlainBinaryDictionary flba(dpage.data(), 48, 12, 4);
    EXPECT(flba.decodeToBinary(2) == std::string((const char*)dpage.data() + 8, 4));
    EXPECT(throws<std::out_of_range>([&] { PlainBinaryDictionary cut(dpage.data(), dpage.size() - 9, 50); }));
  }

  // ByteStreamSplitValuesReaderTest: floats, element bytes spread over K streams
  {
    const float vals[3] = {2.0f, -1.5f, NAN};
    uint8_t raw[12];
    std::memcpy(raw, vals, 12);
    raw[8] = 0x01;  // a NaN payload must survive
    std::vector<uint8_t> page(12);
    for (int i = 0; i < 3; ++i)
      for (int s = 0; s < 4; ++s) page[s * 3 + i] = raw[i * 4 + s];
    ByteBufferInputStream in = ByteBufferInputStream::wrap(page);
    ByteStreamSplitValuesReader r(4);
    r.initFromPage(3, in);
    EXPECT(r.readFloat() == 2.0f && r.readFloat() == -1.5f);
    EXPECT(std::memcmp(r.values() + 8, raw + 8, 4) == 0);
    r.skip(1);
    EXPECT(throws<ParquetDecodingException>([&] { r.readFloat(); }));
    std::vector<uint8_t> odd(7, 0);
    ByteBufferInputStream in2 = ByteBufferInputStream::wrap(odd);
    ByteStreamSplitValuesReader r2(4);
    EXPECT(throws<ParquetDecodingException>([&] { r2.initFromPage(2, in2); }));
  }

  // DeltaBinaryPackingValuesReader: block 128 / 4 miniblocks, first value 7, then constant delta 3
  // (header: varints 128, 4, count, zigzag(first); block: zigzag(minDelta=3), widths 0 0 0 0)
  {
    const int n = 100;
    std::vector<uint8_t> page = {0x80, 0x01, 4, (uint8_t)n, 14, 6, 0, 0, 0, 0};
    page.push_back(0x55);  // next section
    ByteBufferInputStream in = ByteBufferInputStream::wrap(page);
    DeltaBinaryPackingValuesReader r(8);
    r.initFromPage(n, in);
    EXPECT(r.getNextOffset() == 10 && in.position() == 10);
    bool ok = true;
    for (int i = 0; i < n; ++i) ok = ok && r.readLong() == 7 + 3LL * i;
    EXPECT(ok);
    EXPECT(throws<ParquetDecodingException>([&] { r.readLong(); }));
  }

  // legacy BIT_PACKED levels, MSB first (TestBitPackingColumn: 0..7 at width 3 = 00000101 00111001 01110111)
  {
    std::vector<uint8_t> page = {0x05, 0x39, 0x77};
    ByteBufferInputStream in = ByteBufferInputStream::wrap(page);
    ByteBitPackingValuesReader r(7);
    r.initFromPage(8, in);
    for (int i = 0; i < 8; ++i) EXPECT(r.readInteger() == i);
    EXPECT(r.getNextOffset() == 3);
  }

  std::printf(failures ? "%d FAILURES\n" : "all reader-mirror checks passed\n", failures);
  return failures ? 1 : 0;
}
