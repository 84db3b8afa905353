/* CPU restatement of GZIP page decompression.  TEST INFRASTRUCTURE: only tests/, the smoke check and the
 * benchmark's CPU baseline use it; the product library never does.
 *
 * parquet-java hands a GZIP page to Hadoop's GzipCodec (CodecFactory.HeapBytesDecompressor.decompress,
 * parquet-hadoop/src/main/java/org/apache/parquet/hadoop/CodecFactory.java:165-217; the codec class is named in
 * CompressionCodecName.GZIP, parquet-common/.../hadoop/metadata/CompressionCodecName.java:28), whose
 * decompressors (zlib through JNI, or org.apache.hadoop.io.compress.zlib.BuiltInGzipDecompressor) are not in
 * the reference tree.  The formats are public: RFC 1952 (the gzip member: header, deflate data, CRC-32 and
 * ISIZE) and RFC 1951 (deflate: stored, fixed and dynamic Huffman blocks).  This file restates them; the
 * block decoding follows the structure of the format's own reference decoder (canonical codes by count /
 * offset tables, a code set may be incomplete only when it is a single code of length one).
 *
 * Rules of this restatement (the HIP kernel follows the same ones, so that the two agree on every input):
 *  - a page body is one or more complete members; every member's CRC-32 and ISIZE are checked;
 *  - the members together must produce exactly `cap` bytes (the page header's uncompressed size);
 *  - a symbol's bits must lie inside the stream: the decoder fails at the first symbol (or header field)
 *    that needs a bit behind the last byte;
 *  - on failure: ORC_DECODE, *produced = the bytes of all symbols in front of the failing one, *consumed = 0.
 * The JVM path reads exactly uncompressed_page_size bytes from the codec's stream and may never look at
 * the last member's trailer; this restatement always does.  Parity status: pinned by streams the zlib
 * library produced (tests/golden/codec_vectors.json, compressed pyarrow pages) and hand-made streams.
 */
#include <string.h>

#include "pqd_oracle.h"

typedef struct {
  const uint8_t* in;
  uint64_t end_bits; /* 8 * in_len */
  uint64_t bp;       /* next bit */
  uint8_t* out;
  size_t cap, op;
} gz_state;

/* `n` bits at the cursor, least significant first (n <= 32); bits behind the stream read as zero. */
static uint32_t gz_peek(const gz_state* s, unsigned n) {
  uint64_t v = 0;
  uint64_t byte = s->bp >> 3;
  for (unsigned k = 0; k < 6; ++k) {
    uint64_t b = byte + k;
    v |= (uint64_t)(b * 8 < s->end_bits ? s->in[b] : 0) << (8 * k);
  }
  v >>= (s->bp & 7);
  return n >= 32 ? (uint32_t)v : (uint32_t)(v & ((1ull << n) - 1));
}

/* takes n bits; 0 when they lie inside the stream */
static int gz_bits(gz_state* s, unsigned n, uint32_t* v) {
  *v = gz_peek(s, n);
  s->bp += n;
  return s->bp > s->end_bits;
}

#define GZ_MAXBITS 15
typedef struct {
  uint16_t count[GZ_MAXBITS + 1]; /* codes of each length */
  uint16_t symbol[288];           /* symbols ordered by code */
} gz_huff;

/* canonical code from code lengths: 0 complete, < 0 over-subscribed, > 0 incomplete */
static int gz_construct(gz_huff* h, const uint8_t* length, int n) {
  uint16_t offs[GZ_MAXBITS + 1];
  memset(h->count, 0, sizeof h->count);
  for (int i = 0; i < n; ++i) h->count[length[i]]++;
  if (h->count[0] == n) return 0; /* no codes: complete, but decoding fails */
  int left = 1;
  for (int len = 1; len <= GZ_MAXBITS; ++len) {
    left <<= 1;
    left -= h->count[len];
    if (left < 0) return left;
  }
  offs[1] = 0;
  for (int len = 1; len < GZ_MAXBITS; ++len) offs[len + 1] = (uint16_t)(offs[len] + h->count[len]);
  for (int i = 0; i < n; ++i)
    if (length[i]) h->symbol[offs[length[i]]++] = (uint16_t)i;
  return left;
}

/* one symbol: -1 when no code matches the next 15 bits or the code's bits lie behind the stream */
static int gz_decode(gz_state* s, const gz_huff* h) {
  uint32_t bits = gz_peek(s, GZ_MAXBITS);
  int code = 0, first = 0, index = 0;
  for (int len = 1; len <= GZ_MAXBITS; ++len) {
    code |= (int)(bits & 1);
    bits >>= 1;
    int count = h->count[len];
    if (code - count < first) {
      s->bp += (unsigned)len;
      if (s->bp > s->end_bits) return -1;
      return h->symbol[index + (code - first)];
    }
    index += count;
    first += count;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

static const uint16_t GZ_LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint16_t GZ_LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t GZ_DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint16_t GZ_DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

/* the symbols of one block up to its end-of-block code; `floor`: the member's first output byte (a distance
 * may not reach in front of it: every member is a deflate stream of its own) */
static int gz_codes(gz_state* s, const gz_huff* lencode, const gz_huff* distcode, size_t floor) {
  for (;;) {
    int sym = gz_decode(s, lencode);
    if (sym < 0) return ORC_DECODE;
    if (sym < 256) {
      if (s->op >= s->cap) return ORC_DECODE;
      s->out[s->op++] = (uint8_t)sym;
    } else if (sym == 256) {
      return ORC_OK;
    } else {
      sym -= 257;
      if (sym >= 29) return ORC_DECODE;
      uint32_t extra;
      if (gz_bits(s, GZ_LEXT[sym], &extra)) return ORC_DECODE;
      size_t len = GZ_LBASE[sym] + extra;
      int dsym = gz_decode(s, distcode);
      if (dsym < 0 || dsym >= 30) return ORC_DECODE;
      if (gz_bits(s, GZ_DEXT[dsym], &extra)) return ORC_DECODE;
      size_t dist = GZ_DBASE[dsym] + extra;
      if (dist > s->op - floor) return ORC_DECODE; /* too far back */
      if (len > s->cap - s->op) return ORC_DECODE;
      for (size_t i = 0; i < len; ++i, ++s->op) s->out[s->op] = s->out[s->op - dist];
    }
  }
}

static int gz_inflate(gz_state* s) {
  const size_t floor = s->op;
  for (;;) {
    uint32_t last, type;
    if (gz_bits(s, 1, &last) || gz_bits(s, 2, &type)) return ORC_DECODE;
    if (type == 0) { /* stored: to the next byte, LEN, NLEN, the bytes */
      s->bp = (s->bp + 7) & ~(uint64_t)7;
      uint32_t len, nlen;
      if (gz_bits(s, 16, &len) || gz_bits(s, 16, &nlen)) return ORC_DECODE;
      if ((len ^ 0xFFFFu) != nlen) return ORC_DECODE;
      if (s->bp + 8ull * len > s->end_bits) return ORC_DECODE;
      if (len > s->cap - s->op) return ORC_DECODE;
      memcpy(s->out + s->op, s->in + (s->bp >> 3), len);
      s->op += len;
      s->bp += 8ull * len;
    } else if (type == 1) { /* fixed codes */
      uint8_t lengths[288];
      gz_huff lencode, distcode;
      int i = 0;
      for (; i < 144; ++i) lengths[i] = 8;
      for (; i < 256; ++i) lengths[i] = 9;
      for (; i < 280; ++i) lengths[i] = 7;
      for (; i < 288; ++i) lengths[i] = 8;
      gz_construct(&lencode, lengths, 288);
      for (i = 0; i < 30; ++i) lengths[i] = 5;
      gz_construct(&distcode, lengths, 30);
      int st = gz_codes(s, &lencode, &distcode, floor);
      if (st) return st;
    } else if (type == 2) { /* dynamic codes */
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      uint32_t nlen, ndist, ncode, v;
      uint8_t lengths[320];
      gz_huff lencode, distcode;
      if (gz_bits(s, 5, &nlen) || gz_bits(s, 5, &ndist) || gz_bits(s, 4, &ncode)) return ORC_DECODE;
      nlen += 257;
      ndist += 1;
      ncode += 4;
      if (nlen > 286 || ndist > 30) return ORC_DECODE;
      memset(lengths, 0, sizeof lengths);
      for (uint32_t i = 0; i < ncode; ++i) {
        if (gz_bits(s, 3, &v)) return ORC_DECODE;
        lengths[order[i]] = (uint8_t)v;
      }
      if (gz_construct(&lencode, lengths, 19) != 0) return ORC_DECODE; /* the code-length code must be complete */
      uint32_t index = 0;
      while (index < nlen + ndist) {
        int sym = gz_decode(s, &lencode);
        if (sym < 0) return ORC_DECODE;
        if (sym < 16) {
          lengths[index++] = (uint8_t)sym;
        } else {
          uint32_t len = 0, rep;
          if (sym == 16) {
            if (index == 0) return ORC_DECODE; /* nothing to repeat */
            len = lengths[index - 1];
            if (gz_bits(s, 2, &rep)) return ORC_DECODE;
            rep += 3;
          } else if (sym == 17) {
            if (gz_bits(s, 3, &rep)) return ORC_DECODE;
            rep += 3;
          } else {
            if (gz_bits(s, 7, &rep)) return ORC_DECODE;
            rep += 11;
          }
          if (index + rep > nlen + ndist) return ORC_DECODE;
          while (rep--) lengths[index++] = (uint8_t)len;
        }
      }
      if (lengths[256] == 0) return ORC_DECODE; /* no end-of-block code */
      int err = gz_construct(&lencode, lengths, (int)nlen);
      if (err && (err < 0 || nlen != (uint32_t)(lencode.count[0] + lencode.count[1]))) return ORC_DECODE;
      err = gz_construct(&distcode, lengths + nlen, (int)ndist);
      if (err && (err < 0 || ndist != (uint32_t)(distcode.count[0] + distcode.count[1]))) return ORC_DECODE;
      int st = gz_codes(s, &lencode, &distcode, floor);
      if (st) return st;
    } else {
      return ORC_DECODE;
    }
    if (last) return ORC_OK;
  }
}

uint32_t orc_crc32(uint32_t crc, const uint8_t* p, size_t n) {
  static uint32_t table[256];
  static int ready = 0;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    ready = 1;
  }
  uint32_t c = ~crc;
  for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
  return ~c;
}

int orc_gzip_decompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t cap, size_t* produced, size_t* consumed) {
  gz_state s;
  s.in = in;
  s.end_bits = (uint64_t)in_len * 8;
  s.bp = 0;
  s.out = out;
  s.cap = cap;
  s.op = 0;
  *produced = *consumed = 0;
  int st = ORC_OK;
  size_t ip = 0;
  if (in_len == 0 && cap == 0) return ORC_OK; /* nothing is read from a page of no bytes */
  do { /* one member (RFC 1952 section 2.3) */
    if (in_len - ip < 10 || in[ip] != 0x1F || in[ip + 1] != 0x8B || in[ip + 2] != 8 || (in[ip + 3] & 0xE0)) {
      st = ORC_DECODE;
      break;
    }
    const unsigned flg = in[ip + 3];
    ip += 10;
    if (flg & 4) { /* FEXTRA */
      if (in_len - ip < 2) {
        st = ORC_DECODE;
        break;
      }
      size_t xlen = (size_t)in[ip] | ((size_t)in[ip + 1] << 8);
      ip += 2;
      if (in_len - ip < xlen) {
        st = ORC_DECODE;
        break;
      }
      ip += xlen;
    }
    for (unsigned bit = 8; bit <= 16 && !st; bit <<= 1) { /* FNAME, FCOMMENT: zero-terminated */
      if (!(flg & bit)) continue;
      while (ip < in_len && in[ip]) ++ip;
      if (ip >= in_len) st = ORC_DECODE;
      ++ip;
    }
    if (st) break;
    if (flg & 2) { /* FHCRC */
      if (in_len - ip < 2) {
        st = ORC_DECODE;
        break;
      }
      ip += 2;
    }
    const size_t member_out = s.op;
    s.bp = (uint64_t)ip * 8;
    if ((st = gz_inflate(&s))) break;
    ip = (size_t)((s.bp + 7) >> 3);
    if (in_len - ip < 8) {
      st = ORC_DECODE;
      break;
    }
    const uint32_t crc = (uint32_t)in[ip] | ((uint32_t)in[ip + 1] << 8) | ((uint32_t)in[ip + 2] << 16) | ((uint32_t)in[ip + 3] << 24);
    const uint32_t isize = (uint32_t)in[ip + 4] | ((uint32_t)in[ip + 5] << 8) | ((uint32_t)in[ip + 6] << 16) | ((uint32_t)in[ip + 7] << 24);
    ip += 8;
    if (isize != (uint32_t)(s.op - member_out) || crc != orc_crc32(0, out + member_out, s.op - member_out)) {
      st = ORC_DECODE;
      break;
    }
  } while (ip < in_len);
  if (st == ORC_OK && s.op != cap) st = ORC_DECODE; /* the page header announced cap bytes */
  *produced = s.op;
  *consumed = st == ORC_OK ? ip : 0;
  return st;
}
