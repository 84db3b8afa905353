// Page decompression on the device: GZIP (RFC 1952 members around RFC 1951 deflate data).
//
// Replaces the GZIP leg of CodecFactory.HeapBytesDecompressor.decompress (parquet-hadoop/.../hadoop/
// CodecFactory.java:165-217: codec.createInputStream(bytes, decompressor), then exactly uncompressed_page_size
// bytes are read; CompressionCodecName.GZIP names org.apache.hadoop.io.compress.GzipCodec,
// parquet-common/.../hadoop/metadata/CompressionCodecName.java:28) as ColumnChunkPageReadStore calls it
// (ColumnChunkPageReadStore.java:151-181, :232-262).  A page body is one or more members: a header (ten bytes and
// the optional FEXTRA / FNAME / FCOMMENT / FHCRC fields), deflate blocks (stored; fixed or dynamic Huffman codes
// for literals / lengths and for distances), CRC-32 and length of the member's bytes.
// One wave per page, like the other codecs (pqd_lz_common.hpp: the input staged through LDS, the last 16 or 8 KiB
// of output in an LDS ring).  Headers, code construction and the symbol decode run wave-uniformly (every lane
// computes the same values); a code of up to nine bits is one look-up in a 512-entry table in LDS, longer codes
// are found length by length from the canonical counts; literals are kept one per lane until 64 are there or
// a match needs them, matches are copied by the 64 lanes.  The CRC-32 of a member is taken over its bytes in
// memory: every lane a slice, the slices' values joined by the operator "append n zero bytes" of the CRC's
// field (the operator is squared up from one bit, as the format's library joins the CRCs of two strings).
// Every member's CRC-32 and length are checked, the members together must produce exactly dst_len bytes, and a
// symbol whose bits do not lie inside the stream fails: PQD_DECODE, values_written = the bytes of the symbols
// in front of it, bytes_consumed = 0.
// Not HBM-bound: a serial chain of symbols per page; throughput is the number of pages in flight.
#include <cstdlib>

#include "pqd_device.hpp"
#include "pqd_kernels.hpp"
#include "pqd_lz_common.hpp"

namespace pqd {

// Bytes of output kept in LDS for the matches (farther ones are read back from memory): 16 KiB while all pages of the
// call are in flight with it (28 KB of LDS a page: five pages a CU), 8 KiB in larger calls (20 KB: eight pages a CU;
// 1792 pages of 1 MiB: 33 -> 59 GB/s on text, 6 -> 11 on small integers).
constexpr uint32_t GZ_RING_FEW = 16384, GZ_RING_MANY = 8192, GZ_FEW_PAGES = 1280;
constexpr uint32_t GZ_FAST_BITS = 9;

struct GzLds {
  uint16_t ltab[1u << GZ_FAST_BITS], dtab[1u << GZ_FAST_BITS];  // codes of up to nine bits: length << 9 | symbol (0: none)
  uint16_t lcount[16], dcount[16];                              // codes of each length
  uint16_t lsym[288], dsym[32];                                 // symbols in the order of their codes
  uint16_t next[16];
  uint8_t lens[320];
  uint32_t crc_table[256];
  // What a literal / length code of up to nine bits stands for, in one word: code bits | extra bits << 4 | kind << 8 |
  // value << 16 (GZ_LITERAL: the byte; GZ_LENGTH: the base length; GZ_END; GZ_BAD: symbols 286 and 287), and the same
  // for distance codes (kind GZ_LENGTH: the base distance; GZ_BAD: symbols 30 and 31).  0: the code is longer.
  uint32_t lfast[1u << GZ_FAST_BITS], dfast[1u << GZ_FAST_BITS];
  uint16_t lbase[32], dbase[32];  // the format's tables of bases and extra bits, out of memory
  uint8_t lext[32], dext[32];
};
constexpr uint32_t GZ_LITERAL = 0u, GZ_LENGTH = 1u, GZ_END = 2u, GZ_BAD = 3u;

__device__ const uint16_t GZ_LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t GZ_LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t GZ_DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t GZ_DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t GZ_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// 32 bits of the stream at bit position bp (least significant first; wave-uniform).  Bits behind the stream's end
// are whatever the staging holds: every use is followed by a check of the position against the end.
__device__ __forceinline__ uint32_t gz_peek(InStage& in, uint64_t bp) {
  const uint32_t byte = (uint32_t)(bp >> 3);
  in.reach(byte);
  return (uint32_t)(in.peek8(byte) >> ((uint32_t)bp & 7u));
}

// The same with the last eight bytes fetched kept in registers: a fetch every 25 to 32 bits instead of one per use.
struct GzBits {
  uint64_t held;     // stream bytes at_bit / 8 .. + 7
  uint64_t at_bit;   // (a multiple of 8; ~0: nothing held)
  __device__ __forceinline__ uint32_t peek(InStage& in, uint64_t bp) {
    uint64_t off = bp - at_bit;
    if (off > 32u) {  // (also when bp lies in front of what is held)
      const uint32_t byte = (uint32_t)(bp >> 3);
      in.reach(byte);
      held = in.peek8(byte);
      at_bit = (uint64_t)byte * 8u;
      off = bp & 7u;
    }
    return (uint32_t)(held >> off);
  }
};

// the table word of literal / length symbol `sym` whose code has `len` bits
__device__ __forceinline__ uint32_t gz_lword(const GzLds& L, uint32_t sym, uint32_t len) {
  if (sym < 256u) return len | (GZ_LITERAL << 8) | (sym << 16);
  if (sym == 256u) return len | (GZ_END << 8);
  if (sym >= 286u) return len | (GZ_BAD << 8);
  return len | ((uint32_t)L.lext[sym - 257u] << 4) | (GZ_LENGTH << 8) | ((uint32_t)L.lbase[sym - 257u] << 16);
}
__device__ __forceinline__ uint32_t gz_dword(const GzLds& L, uint32_t sym, uint32_t len) {
  if (sym >= 30u) return len | (GZ_BAD << 8);
  return len | ((uint32_t)L.dext[sym] << 4) | (GZ_LENGTH << 8) | ((uint32_t)L.dbase[sym] << 16);
}

// Canonical code from the lengths lens[0..n): counts per length, symbols in code order, the look-up table of the codes
// of up to nine bits.  Returns 0: complete, < 0: over-subscribed, > 0: incomplete (as the format's reference decoder).
__device__ __forceinline__ int gz_construct(GzLds& L, uint16_t* count, uint16_t* symtab, uint16_t* tab, const uint8_t* lens, uint32_t n,
                                            uint32_t lane) {
  __syncthreads();
  if (lane < 16) count[lane] = 0;
  for (uint32_t k = lane; k < (1u << GZ_FAST_BITS); k += 64u) tab[k] = 0;
  __syncthreads();
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t l = uni(lens[i]);
    count[l] = (uint16_t)(count[l] + 1u);  // (every lane the same value to the same place)
  }
  __syncthreads();
  if (uni(count[0]) == n) return 0;  // no codes: complete, but decoding fails
  int left = 1;
  for (uint32_t len = 1; len <= 15; ++len) {
    left <<= 1;
    left -= (int)uni(count[len]);
    if (left < 0) return left;
  }
  // where the symbols of each length begin in the symbol table
  {
    uint32_t offs = 0;
    for (uint32_t len = 1; len <= 15; ++len) {
      const uint32_t c = uni(count[len]);
      L.next[len] = (uint16_t)offs;
      offs += c;
    }
  }
  __syncthreads();
  uint32_t first[16];
  {
    uint32_t code = 0;
    first[0] = 0;
#pragma unroll
    for (uint32_t len = 1; len <= 15; ++len) {
      first[len] = code;
      code = (code + uni(count[len])) << 1;
    }
  }
  // symbols in code order; a code of up to nine bits fills its places in the table (bit-reversed: the stream is
  // read least significant bit first, a code's first bit is its most significant)
  uint32_t used[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) used[k] = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t l = uni(lens[i]);
    if (l == 0) continue;
    const uint32_t slot = uni(L.next[l]);
    symtab[slot] = (uint16_t)i;
    L.next[l] = (uint16_t)(slot + 1u);
    if (l <= GZ_FAST_BITS) {
      uint32_t f = 0, u = 0;
#pragma unroll
      for (uint32_t k = 1; k <= GZ_FAST_BITS; ++k) {
        f = k == l ? first[k] : f;
        u = k == l ? used[k] : u;
      }
      const uint32_t code = f + u;
      const uint32_t rev = __builtin_bitreverse32(code) >> (32u - l);
      for (uint32_t k = rev + (lane << l); k < (1u << GZ_FAST_BITS); k += 64u << l) tab[k] = (uint16_t)((l << 9) | i);
    }
#pragma unroll
    for (uint32_t k = 1; k <= 15; ++k) used[k] += k == l ? 1u : 0u;
  }
  __syncthreads();
  return left;
}

// One symbol of the code {count, symtab, tab} at bp; advances bp.  -1: no code matches the next 15 bits.
__device__ __forceinline__ int gz_symbol(InStage& in, uint64_t& bp, const uint16_t* count, const uint16_t* symtab, const uint16_t* tab) {
  uint32_t bits = gz_peek(in, bp);
  const uint32_t e = uni(tab[bits & ((1u << GZ_FAST_BITS) - 1u)]);
  if (e) {
    bp += e >> 9;
    return (int)(e & 511u);
  }
  int code = 0, first = 0, index = 0;
  for (uint32_t len = 1; len <= 15; ++len) {
    code |= (int)(bits & 1u);
    bits >>= 1;
    const int c = (int)uni(count[len]);
    if (code - c < first) {
      bp += len;
      return (int)uni(symtab[index + (code - first)]);
    }
    index += c;
    first += c;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

// The symbol of a code of more than nine bits that starts the 32 bits `bits`; len = its length.  -1: no code matches.
__device__ __forceinline__ int gz_long_symbol(uint32_t bits, const uint16_t* count, const uint16_t* symtab, uint32_t& len_out) {
  int code = 0, first = 0, index = 0;
  for (uint32_t len = 1; len <= 15; ++len) {
    code |= (int)(bits & 1u);
    bits >>= 1;
    const int c = (int)uni(count[len]);
    if (code - c < first) {
      len_out = len;
      return (int)uni(symtab[index + (code - first)]);
    }
    index += c;
    first += c;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

// mat * vec over GF(2): lane n < 32 holds row n of the matrix in `row`
__device__ __forceinline__ uint32_t gz_gf2_times(uint32_t row, uint32_t vec) {
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)row, i);
    sum ^= (vec >> i) & 1u ? r : 0u;
  }
  return sum;
}

// CRC-32 (the one of RFC 1952) of the n bytes at p, which this wave has written and waited for.
__device__ __forceinline__ uint32_t gz_crc32(const PQD_GLOBAL uint8_t* p, uint32_t n, const uint32_t* table, uint32_t lane) {
  if (n == 0) return 0u;
  // slices: the first one takes what is left over, the others are `slice` bytes each
  uint32_t slice = (n + 63u) / 64u;
  if (slice < 64u) slice = 64u;
  const uint32_t k = (n + slice - 1u) / slice;  // slices in use, at most 64
  const uint32_t first_len = n - (k - 1u) * slice;
  const uint32_t start = lane == 0 ? 0u : first_len + (lane - 1u) * slice;
  const uint32_t len = lane >= k ? 0u : (lane == 0 ? first_len : slice);
  uint32_t c = 0xFFFFFFFFu;
  for (uint32_t i = 0; i < len; ++i) {
    const uint32_t b = __hip_atomic_load(p + start + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (past the vector L1, which may hold an older line)
    c = table[(c ^ b) & 0xFFu] ^ (c >> 8);
  }
  c = ~c;
  if (lane >= k) c = 0u;
  if (k == 1u) return (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
  // the operator "append `slice` zero bytes", squared up from the operator for one zero bit; lane n < 32: row n
  uint32_t row = lane == 0 ? 0xEDB88320u : (lane < 32 ? 1u << (lane - 1u) : 0u);
  row = gz_gf2_times(row, row);  // two bits
  row = gz_gf2_times(row, row);  // four
  row = gz_gf2_times(row, row);  // one byte
  uint32_t op_row = lane < 32 ? 1u << lane : 0u;  // identity
  bool have = false;
  for (uint32_t l = slice; l; l >>= 1) {
    if (l & 1u) {
      op_row = have ? gz_gf2_times(row, op_row) : row;  // (rows of the product M * O: M times each row vector of O)
      have = true;
    }
    if (l >> 1) row = gz_gf2_times(row, row);
  }
  // crc(A || B) = O(crc(A)) ^ crc(B), slice by slice
  uint32_t acc = (uint32_t)__builtin_amdgcn_readlane((int)c, 0);
  for (uint32_t i = 1; i < k; ++i) acc = gz_gf2_times(op_row, acc) ^ (uint32_t)__builtin_amdgcn_readlane((int)c, (int)i);
  return acc;
}

template <uint32_t GZ_RING>
__global__ __launch_bounds__(64) void k_gzip_pages(const pqd_decompress_desc* __restrict__ descs, pqd_page_result* __restrict__ results,
                                                   uint32_t n_pages) {
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[GZ_RING];
  __shared__ __attribute__((aligned(16))) uint8_t s_stage[LZ_STAGE];
  __shared__ GzLds L;
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_decompress_desc d = descs[page];
  if (d.codec != (uint32_t)PQD_CODEC_GZIP) return;  // another kernel's page
  const uint32_t lane = threadIdx.x;
  if (d.src_len > 0x0FFFFFFFu) {  // (bit positions are kept in 64 bits, byte positions in 32)
    if (lane == 0) {
      pqd_page_result r = {PQD_BAD_ARG, 0u, 0u, 0u};
      results[page] = r;
    }
    return;
  }
  for (uint32_t i = lane; i < 256u; i += 64u) {
    uint32_t c = i;
#pragma unroll
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    L.crc_table[i] = c;
  }
  if (lane < 32u) {
    L.lbase[lane] = lane < 29u ? GZ_LBASE[lane] : (uint16_t)0;
    L.lext[lane] = lane < 29u ? GZ_LEXT[lane] : (uint8_t)0;
    L.dbase[lane] = lane < 30u ? GZ_DBASE[lane] : (uint16_t)0;
    L.dext[lane] = lane < 30u ? GZ_DEXT[lane] : (uint8_t)0;
  }
  GzBits bits = {0ull, ~0ull};
  InStage in;
  in.base = (const PQD_GLOBAL uint8_t*)((uintptr_t)d.src & ~(uintptr_t)15);
  const uint32_t skew = (uint32_t)((uintptr_t)d.src & 15u);
  in.end = skew + d.src_len;
  in.lane = lane;
  in.lds = s_stage;
  in.cur = 0;
  in.start(0);
  in.win_at = 0x80000000u;
  in.win0 = in.win1 = 0;
  OutRing out = {s_ring, (PQD_GLOBAL uint8_t*)d.dst, lane, GZ_RING - 1u};
  const uint32_t cap = d.dst_len, in_end = in.end;
  const uint64_t end_bits = (uint64_t)in_end * 8u;
  uint32_t ip = skew, op = 0;
  int32_t status = PQD_OK;
  uint32_t mylit = 0, nl = 0;  // literals not yet written: lane i holds the i-th
  uint64_t bp = 0;
  uint32_t member_out = 0;
  bool more = !(d.src_len == 0u && cap == 0u);  // nothing is read from a page of no bytes

#define GZ_FLUSH()                                       \
  {                                                      \
    if (nl) {                                            \
      if (lane < nl) {                                   \
        s_ring[(op + lane) & (GZ_RING - 1u)] = (uint8_t)mylit; \
        out.dst[op + lane] = (uint8_t)mylit;             \
      }                                                  \
      op += nl;                                          \
      nl = 0;                                            \
    }                                                    \
  }
#define GZ_FAIL()          \
  {                        \
    status = PQD_DECODE;   \
    goto done;             \
  }
#define GZ_TAKE(var, nbits)                                               \
  {                                                                       \
    var = gz_peek(in, bp) & ((nbits) >= 32u ? ~0u : (1u << (nbits)) - 1u); \
    bp += (nbits);                                                        \
    if (bp > end_bits) GZ_FAIL();                                         \
  }

  while (more) {
    // ---- member header (RFC 1952 section 2.3)
    {
      if (in_end - ip < 10u) GZ_FAIL();
      in.reach(ip);
      const uint64_t h = in.peek8(ip);
      if ((h & 0xFFu) != 0x1Fu || ((h >> 8) & 0xFFu) != 0x8Bu || ((h >> 16) & 0xFFu) != 8u || ((h >> 24) & 0xE0u)) GZ_FAIL();
      const uint32_t flg = (uint32_t)(h >> 24) & 0xFFu;
      ip += 10u;
      if (flg & 4u) {  // FEXTRA
        if (in_end - ip < 2u) GZ_FAIL();
        in.reach(ip);
        const uint32_t xlen = (uint32_t)in.peek8(ip) & 0xFFFFu;
        ip += 2u;
        if (in_end - ip < xlen) GZ_FAIL();
        ip += xlen;
      }
      for (uint32_t bit = 8u; bit <= 16u; bit <<= 1) {  // FNAME, FCOMMENT: zero-terminated
        if (!(flg & bit)) continue;
        for (;;) {
          if (ip >= in_end) GZ_FAIL();
          in.reach(ip);
          const uint32_t b = in.ubyte(ip);
          ++ip;
          if (b == 0u) break;
        }
      }
      if (flg & 2u) {  // FHCRC
        if (in_end - ip < 2u) GZ_FAIL();
        ip += 2u;
      }
    }
    member_out = op;
    bp = (uint64_t)ip * 8u;
    // ---- deflate blocks
    for (;;) {
      uint32_t last, type;
      GZ_TAKE(last, 1u);
      GZ_TAKE(type, 2u);
      if (type == 0u) {  // stored
        bp = (bp + 7u) & ~(uint64_t)7;
        uint32_t len, nlen;
        GZ_TAKE(len, 16u);
        GZ_TAKE(nlen, 16u);
        if ((len ^ 0xFFFFu) != nlen) GZ_FAIL();
        if (bp + 8ull * len > end_bits) GZ_FAIL();
        GZ_FLUSH();
        if (len > cap - op) GZ_FAIL();
        const uint32_t from = (uint32_t)(bp >> 3);
        in.reach(from);
        out.literal(in, from, op, len);
        op += len;
        bp += 8ull * len;
      } else if (type == 1u || type == 2u) {
        uint32_t nlen = 288u, ndist = 30u;
        __syncthreads();
        if (type == 1u) {
          for (uint32_t i = lane; i < 320u; i += 64u) L.lens[i] = (uint8_t)(i < 144u ? 8u : i < 256u ? 9u : i < 280u ? 7u : i < 288u ? 8u : 5u);
          __syncthreads();
          (void)gz_construct(L, L.lcount, L.lsym, L.ltab, L.lens, 288u, lane);
          (void)gz_construct(L, L.dcount, L.dsym, L.dtab, L.lens + 288, 30u, lane);
        } else {
          uint32_t ncode;
          GZ_TAKE(nlen, 5u);
          GZ_TAKE(ndist, 5u);
          GZ_TAKE(ncode, 4u);
          nlen += 257u;
          ndist += 1u;
          ncode += 4u;
          if (nlen > 286u || ndist > 30u) GZ_FAIL();
          for (uint32_t i = lane; i < 320u; i += 64u) L.lens[i] = 0;
          __syncthreads();
          for (uint32_t i = 0; i < ncode; ++i) {
            uint32_t v;
            GZ_TAKE(v, 3u);
            L.lens[GZ_ORDER[i]] = (uint8_t)v;
          }
          __syncthreads();
          if (gz_construct(L, L.lcount, L.lsym, L.ltab, L.lens, 19u, lane) != 0) GZ_FAIL();  // the code-length code must be complete
          // the lengths of both codes, read with the code-length code into lens[0 .. nlen + ndist): the table they are
          // read with lives in lcount / lsym / ltab, so they are collected behind a barrier-free copy: lens is only
          // written here and only read by the constructions below
          for (uint32_t i = lane; i < 19u; i += 64u) L.lens[i] = 0;  // (the code-length lengths are used up)
          __syncthreads();
          uint32_t index = 0;
          while (index < nlen + ndist) {
            const int sym = gz_symbol(in, bp, L.lcount, L.lsym, L.ltab);
            if (sym < 0 || bp > end_bits) GZ_FAIL();
            if (sym < 16) {
              L.lens[index++] = (uint8_t)sym;
            } else {
              uint32_t len = 0, rep;
              if (sym == 16) {
                if (index == 0u) GZ_FAIL();
                len = uni(L.lens[index - 1u]);
                GZ_TAKE(rep, 2u);
                rep += 3u;
              } else if (sym == 17) {
                GZ_TAKE(rep, 3u);
                rep += 3u;
              } else {
                GZ_TAKE(rep, 7u);
                rep += 11u;
              }
              if (index + rep > nlen + ndist) GZ_FAIL();
              for (uint32_t r = 0; r < rep; ++r) L.lens[index++] = (uint8_t)len;
            }
          }
          __syncthreads();
          if (uni(L.lens[256]) == 0u) GZ_FAIL();  // no end-of-block code
          // the distance lengths move out of the way of the literal / length table's 288 places
          uint8_t keep_d = 0;
          if (lane < 32u) keep_d = lane < ndist ? L.lens[nlen + lane] : (uint8_t)0;
          __syncthreads();
          for (uint32_t i = nlen + lane; i < 320u; i += 64u) L.lens[i] = 0;
          __syncthreads();
          if (lane < 32u) L.lens[288u + lane] = keep_d;
          __syncthreads();
          int err = gz_construct(L, L.lcount, L.lsym, L.ltab, L.lens, nlen, lane);
          if (err && (err < 0 || nlen != (uint32_t)(uni(L.lcount[0]) + uni(L.lcount[1])))) GZ_FAIL();
          err = gz_construct(L, L.dcount, L.dsym, L.dtab, L.lens + 288, ndist, lane);
          if (err && (err < 0 || ndist != (uint32_t)(uni(L.dcount[0]) + uni(L.dcount[1])))) GZ_FAIL();
        }
        // ---- the block's symbols: the two tables in words that hold everything a code stands for
        for (uint32_t k = lane; k < (1u << GZ_FAST_BITS); k += 64u) {
          const uint32_t el = L.ltab[k], ed = L.dtab[k];
          L.lfast[k] = el ? gz_lword(L, el & 511u, el >> 9) : 0u;
          L.dfast[k] = ed ? gz_dword(L, ed & 511u, ed >> 9) : 0u;
        }
        __syncthreads();
        for (;;) {
          const uint32_t b = bits.peek(in, bp);
          uint32_t e = uni(L.lfast[b & ((1u << GZ_FAST_BITS) - 1u)]);
          if (!e) {
            uint32_t l = 0;
            const int sym = gz_long_symbol(b, L.lcount, L.lsym, l);
            if (sym < 0) GZ_FAIL();
            e = gz_lword(L, (uint32_t)sym, l);
          }
          const uint32_t l = e & 15u, kind = (e >> 8) & 3u;
          if (kind == GZ_LITERAL) {
            bp += l;
            if (bp > end_bits) GZ_FAIL();
            if (op + nl >= cap) GZ_FAIL();
            mylit = lane == nl ? e >> 16 : mylit;
            if (++nl == 64u) GZ_FLUSH();
            continue;
          }
          if (kind == GZ_END) {
            bp += l;
            if (bp > end_bits) GZ_FAIL();
            break;
          }
          if (kind == GZ_BAD) GZ_FAIL();
          // a match: the length's extra bits lie in the 32 bits at hand (15 + 5 at most)
          const uint32_t xl = (e >> 4) & 15u;
          const uint32_t len = (e >> 16) + ((b >> l) & ((1u << xl) - 1u));
          bp += l + xl;
          if (bp > end_bits) GZ_FAIL();
          const uint32_t c = bits.peek(in, bp);
          uint32_t f = uni(L.dfast[c & ((1u << GZ_FAST_BITS) - 1u)]);
          if (!f) {
            uint32_t dl = 0;
            const int ds = gz_long_symbol(c, L.dcount, L.dsym, dl);
            if (ds < 0) GZ_FAIL();
            f = gz_dword(L, (uint32_t)ds, dl);
          }
          if (((f >> 8) & 3u) == GZ_BAD) GZ_FAIL();
          const uint32_t dl = f & 15u, xd = (f >> 4) & 15u;  // (15 + 13 bits at most)
          const uint32_t dist = (f >> 16) + ((c >> dl) & ((1u << xd) - 1u));
          bp += dl + xd;
          if (bp > end_bits) GZ_FAIL();
          GZ_FLUSH();
          if (dist > op - member_out) GZ_FAIL();  // too far back
          if (len > cap - op) GZ_FAIL();
          out.match(op, dist, len);
          op += len;
        }
      } else {
        GZ_FAIL();
      }
      if (last) break;
    }
    GZ_FLUSH();
    // ---- trailer: CRC-32 and length of the member's bytes
    ip = (uint32_t)((bp + 7u) >> 3);
    if (in_end - ip < 8u) GZ_FAIL();
    {
      in.reach(ip);
      const uint64_t t = in.peek8(ip);
      ip += 8u;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the member's bytes are in memory
      const uint32_t crc = gz_crc32(out.dst + member_out, op - member_out, L.crc_table, lane);
      if ((uint32_t)(t >> 32) != op - member_out || (uint32_t)t != crc) GZ_FAIL();
    }
    more = ip < in_end;
  }
  if (op != cap) status = PQD_DECODE;  // the page header announced cap bytes
done:
  GZ_FLUSH();
  if (lane == 0) {
    pqd_page_result r = {status, op, status == PQD_OK ? ip - skew : 0u, 0u};
    results[page] = r;
  }
#undef GZ_FLUSH
#undef GZ_FAIL
#undef GZ_TAKE
}

hipError_t launch_gzip_pages(const pqd_decompress_desc* descs, pqd_page_result* results, uint32_t n_pages, hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  const char* env = getenv("PQD_GZIP_RING");  // 8 / 16: fixes the KiB of output kept in LDS (testing)
  const int forced = env ? atoi(env) : 0;
  if (forced == 16 || (forced != 8 && n_pages <= GZ_FEW_PAGES))
    hipLaunchKernelGGL(k_gzip_pages<GZ_RING_FEW>, dim3(n_pages), dim3(64), 0, stream, descs, results, n_pages);
  else
    hipLaunchKernelGGL(k_gzip_pages<GZ_RING_MANY>, dim3(n_pages), dim3(64), 0, stream, descs, results, n_pages);
  return hipGetLastError();
}

}  // namespace pqd
