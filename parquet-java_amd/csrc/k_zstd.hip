// Page decompression on the device: Zstandard frames.
//
// Replaces ZstdDecompressorStream (parquet-hadoop/.../hadoop/codec/ZstdDecompressorStream.java:30-60, a wrapper
// of zstd-jni's ZstdInputStreamNoFinalizer, i.e. of the C library) as ColumnChunkPageReadStore calls it through
// BytesInputDecompressor.decompress (ColumnChunkPageReadStore.java:151-181, :232-262).  The format is RFC 8878:
// frames of raw / RLE / compressed blocks; a compressed block is a literals section (raw, RLE or
// Huffman-coded in one or four streams) and a sequences section (FSE-coded literal-length / offset /
// match-length codes read backwards from the block's end), executed as "copy `ll` literals, then `ml` bytes
// from `offset` back".  No dictionaries (parquet pages carry none).
// One wave per page.  The serial parts (headers, table descriptions, the sequence decode) run wave-uniformly:
// every lane computes the same values from LDS, the 64 lanes together move the bytes.  The four Huffman
// streams of a block are decoded by four lanes, each with its own bit reader, into the end of the page's
// destination (a block's output never catches up with its unread literals there, because the block ends at
// or before the page's end); they are then staged through LDS like the input.  Tables live in LDS for the
// whole frame (repeat modes and treeless literals reuse them).
// Latency-bound like the other codecs (a serial chain per page); throughput is the number of pages in flight.
#include <cstdlib>

#include "pqd_device.hpp"
#include "pqd_kernels.hpp"
#include "pqd_lz_common.hpp"

namespace pqd {

constexpr uint32_t ZS_RING = 8192;
constexpr uint32_t ZS_BLOCK_MAX = 128 * 1024;

struct ZShared {
  // The sequence decoder's entries, one per FSE state: .x = the code's baseline (lengths), .y = bits of the value
  // behind the baseline | bits of the next state << 8 | next-state base << 16
  u32x2 ll[512], of[256], ml[512];
  uint32_t tab32[512];                 // an FSE table as it is built: symbol | bits << 8 | next-state base << 16
  uint16_t huf[2048];                  // Huffman entries: symbol | bits << 8
  int16_t norm[256];
  uint16_t next[256];
  uint8_t weights[256];
  uint32_t wtab[64];                   // FSE table of the Huffman weights (accuracy log <= 6)
  uint32_t log_ll, log_of, log_ml, have_ll, have_of, have_ml, huf_bits, huf_valid;
  uint32_t rep[3];
  int32_t lane_status[4];
  uint32_t ll_code[36], ml_code[53];  // baseline | extra bits << 24 of the length codes (read per sequence: LDS, not memory)
};

__device__ const int16_t Z_LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                             2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__device__ const int16_t Z_ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                             1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__device__ const int16_t Z_OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                             1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
__device__ const uint32_t Z_LL_BASE[36] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,   10,  11,  12,   13,   14,   15,   16,    18,
                                           20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
__device__ const uint8_t Z_LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                          1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__device__ const uint32_t Z_ML_BASE[53] = {3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                           21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41,
                                           43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
__device__ const uint8_t Z_ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                          0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

__device__ __forceinline__ uint32_t zhighbit(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }

// a byte of the input (wave-uniform)
__device__ __forceinline__ uint32_t zbyte(InStage& in, uint32_t pos) {
  in.reach(pos);
  return uni(in.byte(pos));
}
// up to 25 bits at bit `bit` of the input region that starts at `p0` and is `n` bytes long (zeros behind it)
__device__ __forceinline__ uint32_t zfwd(InStage& in, uint32_t p0, uint32_t n, uint32_t bit, uint32_t k) {
  uint64_t v = 0;
  const uint32_t byte = bit >> 3;
  for (uint32_t i = 0; i < 5; ++i)
    if (byte + i < n) v |= (uint64_t)zbyte(in, p0 + byte + i) << (8 * i);
  return (uint32_t)(v >> (bit & 7u)) & ((1u << k) - 1u);
}
// k <= 16 bits whose lowest is bit `lo` (may be negative: zeros) of the n-byte region at p0
__device__ __forceinline__ uint32_t zback_small(InStage& in, uint32_t p0, uint32_t n, int32_t lo, uint32_t k) {
  if (k == 0) return 0;
  const int32_t b0 = lo >> 3;  // floor
  uint32_t v = 0;
  for (int32_t i = 0; i < 3; ++i) {
    const int32_t b = b0 + i;
    if (b >= 0 && b < (int32_t)n) v |= zbyte(in, p0 + (uint32_t)b) << (8 * i);
  }
  return (v >> ((uint32_t)lo & 7u)) & ((1u << k) - 1u);
}

// FSE decoding table from normalized counts (all lanes write the same entries)
__device__ __forceinline__ bool zfse_build(uint32_t* tab, const int16_t* norm, uint16_t* next, uint32_t n_sym, uint32_t log) {
  const uint32_t size = 1u << log;
  int32_t high = (int32_t)size - 1;
  for (uint32_t s = 0; s < n_sym; ++s) {
    if (norm[s] == -1) {
      tab[high--] = s;
      next[s] = 1;
    } else {
      next[s] = (uint16_t)norm[s];
    }
  }
  const uint32_t step = (size >> 1) + (size >> 3) + 3u, mask = size - 1u;
  uint32_t pos = 0;
  for (uint32_t s = 0; s < n_sym; ++s) {
    for (int32_t i = 0; i < norm[s]; ++i) {
      tab[pos] = s;
      do pos = (pos + step) & mask;
      while ((int32_t)pos > high);
    }
  }
  if (pos != 0) return false;
  lz_sync();
  for (uint32_t u = 0; u < size; ++u) {
    const uint32_t s = tab[u] & 0xFFu;
    const uint32_t ns = next[s];  // (the lanes of the wave run in step: all have read before any writes)
    next[s] = (uint16_t)(ns + 1u);
    const uint32_t nb = log - zhighbit(ns);
    tab[u] = s | (nb << 8) | ((((ns << nb) - size) & 0xFFFFu) << 16);
  }
  lz_sync();
  return true;
}

// normalized counts at input position p0 (n bytes available); returns bytes used or ~0u
__device__ __forceinline__ uint32_t zfse_norm(InStage& in, uint32_t p0, uint32_t n, int16_t* norm, uint32_t& n_sym, uint32_t max_sym,
                              uint32_t& log, uint32_t max_log) {
  uint32_t bit = 0;
  const uint32_t al = 5u + zfwd(in, p0, n, bit, 4);
  bit += 4;
  if (al > max_log) return ~0u;
  int32_t remaining = (1 << al) + 1, threshold = 1 << al;
  uint32_t nbits = al + 1, sym = 0;
  bool prev0 = false;
  while (remaining > 1 && sym <= max_sym) {
    if (prev0) {
      for (;;) {
        const uint32_t rep = zfwd(in, p0, n, bit, 2);
        bit += 2;
        for (uint32_t i = 0; i < rep && sym <= max_sym; ++i) norm[sym++] = 0;
        if (rep != 3) break;
      }
      if (sym > max_sym) break;
    }
    const int32_t max = (2 * threshold - 1) - remaining;
    const uint32_t bits = zfwd(in, p0, n, bit, nbits);
    int32_t count;
    if ((int32_t)(bits & (uint32_t)(threshold - 1)) < max) {
      count = (int32_t)(bits & (uint32_t)(threshold - 1));
      bit += nbits - 1;
    } else {
      count = (int32_t)(bits & (uint32_t)(2 * threshold - 1));
      if (count >= threshold) count -= max;
      bit += nbits;
    }
    --count;
    remaining -= count < 0 ? -count : count;
    norm[sym++] = (int16_t)count;
    prev0 = count == 0;
    while (remaining < threshold) {
      --nbits;
      threshold >>= 1;
    }
    if (bit > n * 8u + 32u) return ~0u;
  }
  if (remaining != 1 || sym > max_sym + 1) return ~0u;
  n_sym = sym;
  log = al;
  const uint32_t used = (bit + 7u) >> 3;
  return used > n ? ~0u : used;
}

// Huffman table from the weights in sh.weights[0 .. n_given): false when they do not describe a tree
__device__ __forceinline__ bool zhuf_build(ZShared& sh, uint32_t n_given) {
  uint32_t total = 0;
  for (uint32_t i = 0; i < n_given; ++i) {
    const uint32_t w = sh.weights[i];
    if (w > 11) return false;
    total += w ? 1u << (w - 1) : 0u;
  }
  if (total == 0) return false;
  const uint32_t max_bits = zhighbit(total) + 1u;
  if (max_bits > 11) return false;
  const uint32_t rest = (1u << max_bits) - total;
  if (rest & (rest - 1u)) return false;
  sh.weights[n_given] = (uint8_t)(zhighbit(rest) + 1u);
  lz_sync();
  const uint32_t n = n_given + 1;
  uint32_t start[13], cnt[13];
  for (int r = 0; r < 13; ++r) start[r] = cnt[r] = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t w = sh.weights[i];
    for (int r = 0; r < 13; ++r) cnt[r] += (uint32_t)r == w ? 1u : 0u;
  }
  uint32_t at = 0;
  for (uint32_t r = 1; r <= max_bits; ++r) {
    start[r] = at;
    at += cnt[r] << (r - 1);
  }
  if (at != (1u << max_bits)) return false;
  // symbols of one weight lie one behind the other in symbol order; the 64 lanes fill an entry range together
  const uint32_t lane = threadIdx.x;
  for (uint32_t s = 0; s < n; ++s) {
    const uint32_t w = sh.weights[s];
    if (!w) continue;
    const uint32_t len = 1u << (w - 1);
    uint32_t first = 0;
    for (int r = 0; r < 13; ++r) first = (uint32_t)r == w ? start[r] : first;
    const uint16_t e = (uint16_t)(s | ((max_bits + 1u - w) << 8));
    for (uint32_t i = lane; i < len; i += 64u) sh.huf[first + i] = e;
    for (int r = 0; r < 13; ++r) start[r] += (uint32_t)r == w ? len : 0u;
  }
  sh.huf_bits = max_bits;
  sh.huf_valid = 1;
  lz_sync();
  return true;
}

// Huffman tree description at input position p0 (n bytes available); returns bytes used or ~0u
__device__ __forceinline__ uint32_t zhuf_tree(ZShared& sh, InStage& in, uint32_t p0, uint32_t n) {
  if (n == 0) return ~0u;
  const uint32_t hb = zbyte(in, p0);
  uint32_t n_w = 0, used;
  if (hb >= 128) {
    n_w = hb - 127;
    used = 1 + (n_w + 1) / 2;
    if (used > n) return ~0u;
    for (uint32_t i = 0; i < n_w; ++i) {
      const uint32_t b = zbyte(in, p0 + 1 + i / 2);
      sh.weights[i] = (uint8_t)((i & 1u) ? b & 15u : b >> 4);
    }
  } else {
    used = 1 + hb;
    if (used > n || hb == 0) return ~0u;
    uint32_t n_sym, log;
    const uint32_t tb = zfse_norm(in, p0 + 1, hb, sh.norm, n_sym, 255, log, 6);
    if (tb == ~0u) return ~0u;
    lz_sync();
    if (!zfse_build(sh.wtab, sh.norm, sh.next, n_sym, log)) return ~0u;
    const uint32_t sp = p0 + 1 + tb, sn = hb - tb;
    if (sn == 0) return ~0u;
    const uint32_t last = zbyte(in, sp + sn - 1);
    if (last == 0) return ~0u;
    int32_t bit = (int32_t)(sn - 1) * 8 + (int32_t)zhighbit(last);
    bit -= (int32_t)log;
    uint32_t s1 = zback_small(in, sp, sn, bit, log);
    bit -= (int32_t)log;
    uint32_t s2 = zback_small(in, sp, sn, bit, log);
    for (;;) {
      if (n_w >= 254) return ~0u;
      uint32_t e1 = sh.wtab[s1], e2 = sh.wtab[s2];
      sh.weights[n_w++] = (uint8_t)e1;
      uint32_t nb = (e1 >> 8) & 0xFFu;
      if (bit < (int32_t)nb) {
        if (n_w >= 255) return ~0u;
        sh.weights[n_w++] = (uint8_t)e2;
        break;
      }
      bit -= (int32_t)nb;
      s1 = (e1 >> 16) + zback_small(in, sp, sn, bit, nb);
      sh.weights[n_w++] = (uint8_t)e2;
      nb = (e2 >> 8) & 0xFFu;
      if (bit < (int32_t)nb) {
        if (n_w >= 255) return ~0u;
        sh.weights[n_w++] = (uint8_t)sh.wtab[s1];
        break;
      }
      bit -= (int32_t)nb;
      s2 = (e2 >> 16) + zback_small(in, sp, sn, bit, nb);
    }
  }
  lz_sync();
  if (!zhuf_build(sh, n_w)) return ~0u;
  return used;
}

// One Huffman stream by one lane: `count` symbols from the n bytes at `p` to `out`; PQD_OK when the stream is
// used up exactly.  (Reads 8 bytes at a time from memory; a read in front of the stream supplies zeros.)
__device__ __forceinline__ int32_t zhuf_stream(const ZShared& sh, const PQD_GLOBAL uint8_t* p, uint32_t n, PQD_GLOBAL uint8_t* out, uint32_t count) {
  if (n == 0 || p[n - 1] == 0) return PQD_DECODE;
  int64_t bit = (int64_t)(n - 1) * 8 + (int64_t)zhighbit(p[n - 1]);
  const uint32_t mb = sh.huf_bits;
  uint64_t cache = 0;
  int64_t cache_lo = bit + 1;  // nothing cached
  for (uint32_t i = 0; i < count; ++i) {
    const int64_t lo = bit - (int64_t)mb;  // lowest bit of the peeked field
    if (lo < cache_lo) {
      // 64 bits from a byte boundary, ending at or just above the field's top: at least 46 bits of fields per fill
      const int64_t byte0 = (lo + (int64_t)mb - 64 + 7) >> 3;  // floor
      cache_lo = byte0 * 8;
      if (byte0 >= 0 && byte0 + 8 <= (int64_t)n) {
        __builtin_memcpy(&cache, p + byte0, 8);
      } else {
        cache = 0;
        for (int k = 0; k < 8; ++k) {
          const int64_t q = byte0 + k;
          if (q >= 0 && q < (int64_t)n) cache |= (uint64_t)p[q] << (8 * k);
        }
      }
    }
    const uint32_t v = (uint32_t)(cache >> (lo - cache_lo)) & ((1u << mb) - 1u);
    const uint32_t e = sh.huf[v];
    out[i] = (uint8_t)e;
    bit -= (int64_t)(e >> 8);
  }
  return bit == 0 ? PQD_OK : PQD_DECODE;
}

// The sequences' bit stream, read backwards.  The lanes hold 64 aligned words of the input around the reading
// position (lane l: word `base` + l), and the 64 below them with four words in common, requested when the upper ones
// come into use: a field is two lane reads and a shift, nothing waits for memory or LDS.  Bit positions count from the
// first bit of the word that holds the region's first byte.
struct BackWords {
  const PQD_GLOBAL uint32_t* words;  // the input from its 16-byte aligned base
  int32_t n_words;                   // words that hold bytes of the input
  int32_t w0;                        // the word of the region's first byte
  int32_t base;
  uint32_t cur, nxt;
  uint32_t lane;

  __device__ __forceinline__ uint32_t load(int32_t b) const {
    const int32_t w = w0 + b + (int32_t)lane;
    return w >= 0 && w < n_words ? words[w] : 0u;
  }
  __device__ __forceinline__ void start(int32_t bitpos) {
    base = (bitpos >> 5) - 62;  // (the word above the one a field starts in is read as well: it is a lane of `cur` too)
    cur = load(base);
    nxt = load(base - 60);
  }
  // before a sequence is read at `bitpos`: its fields are 89 bits at most
  __device__ __forceinline__ void need(int32_t bitpos) {
    if ((bitpos >> 5) - 3 < base) {
      base -= 60;
      cur = nxt;
      nxt = load(base - 60);
    }
  }
  // the 32 bits from bit `lo` up
  __device__ __forceinline__ uint32_t raw(int32_t lo) const {
    const uint32_t w = (uint32_t)((lo >> 5) - base);
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)w);
    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(w + 1u));
    return (uint32_t)((((uint64_t)b << 32) | a) >> ((uint32_t)lo & 31u));
  }
  // the k <= 31 bits from bit `lo` up
  __device__ __forceinline__ uint32_t field(int32_t lo, uint32_t k) const { return raw(lo) & ((1u << k) - 1u); }
};

// one table of the sequences section (kind 0: literal lengths, 1: offsets, 2: match lengths); returns bytes used or ~0u
__device__ __forceinline__ uint32_t zseq_table(ZShared& sh, InStage& in, u32x2* tab, uint32_t kind, uint32_t& tlog, uint32_t& have, uint32_t mode,
                               uint32_t p0, uint32_t n, const PQD_GLOBAL int16_t* def, uint32_t def_n, uint32_t def_log, uint32_t max_sym,
                               uint32_t max_log) {
  if (mode == 3) return have ? 0u : ~0u;  // repeat
  uint32_t used, log;
  if (mode == 1) {  // one symbol, no bits
    if (n < 1) return ~0u;
    const uint32_t s = zbyte(in, p0);
    if (s > max_sym) return ~0u;
    lz_sync();
    sh.tab32[0] = s;
    log = 0;
    used = 1;
  } else {
    uint32_t n_sym = def_n;
    log = def_log;
    used = 0;
    if (mode == 0) {
      for (uint32_t i = 0; i < def_n; ++i) sh.norm[i] = def[i];
    } else {
      used = zfse_norm(in, p0, n, sh.norm, n_sym, max_sym, log, max_log);
      if (used == ~0u) return ~0u;
    }
    lz_sync();
    if (!zfse_build(sh.tab32, sh.norm, sh.next, n_sym, log)) return ~0u;
  }
  lz_sync();
  // (symbols are max_sym at most: the description cannot hold more)
  for (uint32_t u = in.lane; u < (1u << log); u += 64u) {
    const uint32_t e = sh.tab32[u], sym = e & 0xFFu;
    const uint32_t code = kind == 1u ? sym << 24 : kind == 0u ? sh.ll_code[sym] : sh.ml_code[sym];
    tab[u] = u32x2{code & 0xFFFFFFu, (e & 0xFFFFFF00u) | (code >> 24)};
  }
  lz_sync();
  tlog = log;
  have = 1;
  return used;
}

// XXH64 of n bytes this wave stored at p (the frame's content checksum is its low 32 bits): the four
// accumulators run in four lanes, the rest is computed by every lane alike.
__device__ __forceinline__ uint64_t zxxh64(const PQD_GLOBAL uint8_t* p, uint32_t n, uint32_t lane) {
  constexpr uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull,
                     P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;
  auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
  auto round1 = [&](uint64_t acc, uint64_t v) { return rotl(acc + v * P2, 31) * P1; };
  auto rd = [&](uint32_t at, uint32_t k) {  // k bytes, little endian
    uint64_t v = 0;
    for (uint32_t i = 0; i < k; ++i) v |= (uint64_t)p[at + i] << (8 * i);
    return v;
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  uint32_t at = 0;
  uint64_t h;
  if (n >= 32u) {
    const uint32_t q = lane & 3u;
    uint64_t v = q == 0 ? P1 + P2 : q == 1 ? P2 : q == 2 ? 0ull : 0ull - P1;
    const uint32_t stripes = n / 32u;
    for (uint32_t s = 0; s < stripes; ++s) {
      uint64_t w;
      __builtin_memcpy(&w, p + (size_t)s * 32u + q * 8u, 8);
      v = round1(v, w);
    }
    at = stripes * 32u;
    uint64_t acc[4];
    for (int k = 0; k < 4; ++k)
      acc[k] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), k) << 32) |
               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, k);
    h = rotl(acc[0], 1) + rotl(acc[1], 7) + rotl(acc[2], 12) + rotl(acc[3], 18);
    for (int k = 0; k < 4; ++k) h = (h ^ round1(0, acc[k])) * P1 + P4;
  } else {
    h = P5;
  }
  h += (uint64_t)n;
  for (; at + 8u <= n; at += 8u) {
    h ^= round1(0, rd(at, 8));
    h = rotl(h, 27) * P1 + P4;
  }
  if (at + 4u <= n) {
    h ^= rd(at, 4) * P1;
    h = rotl(h, 23) * P2 + P3;
    at += 4u;
  }
  for (; at < n; ++at) {
    h ^= rd(at, 1) * P5;
    h = rotl(h, 11) * P1;
  }
  h ^= h >> 33;
  h *= P2;
  h ^= h >> 29;
  h *= P3;
  h ^= h >> 32;
  return h;
}

// ---- where a page's bytes are moved.  The page's serial part (headers, tables, the sequences' bit stream) hands over
// "copy n literals", "repeat ml bytes from offset back", ... : in a call of many pages the same wave carries them out at
// once (ZDirect); in a call of few pages a second wave of the workgroup does (ZPush -> queue in LDS -> zstd_consume), so
// that moving the bytes of a sequence runs beside the decode of the next ones.
struct ZExec {
  OutRing out;
  InStage lits;
  const PQD_GLOBAL uint8_t* in_base;  // the input from its aligned base (raw blocks)
  uint32_t op, lit_pos, lt, rle_byte, lane;

  __device__ __forceinline__ void init(uint8_t* ring, uint32_t ring_bytes, uint8_t* lit_stage, const pqd_decompress_desc& d, uint32_t lane_) {
    out = OutRing{ring, (PQD_GLOBAL uint8_t*)d.dst, lane_, ring_bytes - 1u};
    in_base = (const PQD_GLOBAL uint8_t*)((uintptr_t)d.src & ~(uintptr_t)15);
    lane = lane_;
    lits.lane = lane_;
    lits.lds = lit_stage;
    op = lit_pos = lt = rle_byte = 0;
  }
  // the literals of the block that begins: from memory (lt 0: the input; 2 / 3: decoded by the other side, which has
  // waited for its stores) or one byte repeated (lt 1)
  __device__ __forceinline__ void set_literals(uint32_t lt_, const PQD_GLOBAL uint8_t* base, uint32_t pos, uint32_t end, uint32_t rle) {
    lt = lt_;
    rle_byte = rle;
    lit_pos = pos;
    if (lt_ != 1u) {
      if (lt_ >= 2u) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      lits.base = base;
      lits.end = end;
      lits.cur = 0;
      lits.win_at = 0x80000000u;
      lits.win0 = lits.win1 = 0;
      lits.start(pos / LZ_CHUNK);
    }
  }
  __device__ __forceinline__ void literals(uint32_t n) {
    if (lt == 1u) {
      for (uint32_t i = lane; i < n; i += 64u) {
        out.ring[(op + i) & out.ring_mask] = (uint8_t)rle_byte;
        out.dst[op + i] = (uint8_t)rle_byte;
      }
    } else {
      for (uint32_t done = 0; done < n;) {
        const uint32_t nn = n - done < 1900u ? n - done : 1900u;
        lits.reach(lit_pos + done);
        out.literal(lits, lit_pos + done, op + done, nn, 1900u);
        done += nn;
      }
    }
    lit_pos += n;
    op += n;
  }
  __device__ __forceinline__ void match(uint32_t offset, uint32_t ml) {
    out.match(op, offset, ml);
    op += ml;
  }
  __device__ __forceinline__ void raw(uint32_t p, uint32_t size) {  // a raw block: input position p
    const PQD_GLOBAL uint8_t* src = in_base + p;
#pragma unroll 4
    for (uint32_t i = lane; i < size; i += 64u) {
      const uint8_t v = src[i];
      out.ring[(op + i) & out.ring_mask] = v;
      out.dst[op + i] = v;
    }
    op += size;
  }
  __device__ __forceinline__ void rle_block(uint32_t byte, uint32_t size) {
    for (uint32_t i = lane; i < size; i += 64u) {
      out.ring[(op + i) & out.ring_mask] = (uint8_t)byte;
      out.dst[op + i] = (uint8_t)byte;
    }
    op += size;
  }
  // the frame's content checksum: the low 32 bits of XXH64 of what the frame produced
  __device__ __forceinline__ bool frame_end(uint32_t frame_start, uint32_t want) {
    return uni((uint32_t)zxxh64(out.dst + frame_start, op - frame_start, lane)) == want;
  }
};

struct ZDirect {
  static constexpr bool PIPELINED = false;
  ZExec& x;
  __device__ __forceinline__ void set_literals(uint32_t lt, const PQD_GLOBAL uint8_t* base, uint32_t pos, uint32_t end, uint32_t rle) {
    x.set_literals(lt, base, pos, end, rle);
  }
  __device__ __forceinline__ void literals(uint32_t n) { x.literals(n); }
  __device__ __forceinline__ void match(uint32_t offset, uint32_t ml) { x.match(offset, ml); }
  __device__ __forceinline__ void sequence(uint32_t ll, uint32_t offset, uint32_t ml) {
    if (ll) x.literals(ll);
    x.match(offset, ml);
  }
  __device__ __forceinline__ void raw(uint32_t p, uint32_t size) { x.raw(p, size); }
  __device__ __forceinline__ void rle_block(uint32_t byte, uint32_t size) { x.rle_block(byte, size); }
  __device__ __forceinline__ bool frame_end(uint32_t frame_start, uint32_t want) { return x.frame_end(frame_start, want); }
  __device__ __forceinline__ void drain() {}
  __device__ __forceinline__ bool finish() { return true; }
};

constexpr uint32_t ZQ_N = 256;  // commands in flight
constexpr uint32_t ZC_SEQ = 0, ZC_LITS = 1, ZC_SETLIT = 2, ZC_SETLIT2 = 3, ZC_RAW = 4, ZC_RLE = 5, ZC_FRAME = 6, ZC_DONE = 7;
struct ZQueue {
  u32x4 e[ZQ_N];
  uint32_t head;     // commands written (the producer's)
  uint32_t tail;     // commands taken (the consumer's): their slots are free
  uint32_t done;     // commands carried out
  uint32_t verdict;  // of the last checksum: 1 = fine
  uint32_t abort;    // a side has waited for the other beyond ZQ_PATIENCE: both give the page up (it reports PQD_DECODE)
};
// Rounds of waiting (128 cycles each) a side puts up with: about a second.  Nothing known makes a side wait that long; the
// limit is there so that a mistake ends as a failed page, not as waves that never finish.
constexpr uint32_t ZQ_PATIENCE = 1u << 24;
__device__ __forceinline__ uint32_t zq_load(const uint32_t* p) { return uni(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ __forceinline__ void zq_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// The producer's side (every lane of its wave alike; the other wave is resident in the same workgroup and takes commands
// until it meets ZC_DONE, so a full queue empties and an awaited count is reached).
struct ZPush {
  static constexpr bool PIPELINED = true;
  ZQueue& q;
  uint32_t head, tail_seen;
  bool aborted;
  __device__ __forceinline__ void push(u32x4 e) {
    uint32_t rounds = 0;
    while (head - tail_seen >= ZQ_N && !aborted) {
      tail_seen = zq_load(&q.tail);
      if (head - tail_seen >= ZQ_N) {
        __builtin_amdgcn_s_sleep(2);
        if (++rounds > ZQ_PATIENCE || zq_load(&q.abort)) give_up();
      }
    }
    if (aborted) return;
    q.e[head & (ZQ_N - 1u)] = e;
    ++head;
    // (the count is made known every eighth command: making it known waits for the wave's LDS traffic)
    if ((head & 7u) == 0u) zq_store(&q.head, head);
  }
  __device__ __forceinline__ void give_up() {
    aborted = true;
    zq_store(&q.abort, 1u);
  }
  __device__ __forceinline__ void drain() {  // until everything handed over is carried out
    zq_store(&q.head, head);
    uint32_t rounds = 0;
    while (!aborted && zq_load(&q.done) != head) {
      __builtin_amdgcn_s_sleep(2);
      if (++rounds > ZQ_PATIENCE || zq_load(&q.abort)) give_up();
    }
  }
  __device__ __forceinline__ void set_literals(uint32_t lt, const PQD_GLOBAL uint8_t* base, uint32_t pos, uint32_t end, uint32_t rle) {
    const uint64_t b = (uint64_t)(uintptr_t)base;
    push(u32x4{ZC_SETLIT | (lt << 3) | (rle << 8), (uint32_t)b, (uint32_t)(b >> 32), pos});
    push(u32x4{ZC_SETLIT2, end, 0u, 0u});
  }
  __device__ __forceinline__ void literals(uint32_t n) {
    if (n) push(u32x4{ZC_LITS, n, 0u, 0u});
  }
  __device__ __forceinline__ void match(uint32_t offset, uint32_t ml) { push(u32x4{ZC_SEQ, 0u, ml, offset}); }
  __device__ __forceinline__ void sequence(uint32_t ll, uint32_t offset, uint32_t ml) { push(u32x4{ZC_SEQ, ll, ml, offset}); }
  __device__ __forceinline__ void raw(uint32_t p, uint32_t size) { push(u32x4{ZC_RAW, size, p, 0u}); }
  __device__ __forceinline__ void rle_block(uint32_t byte, uint32_t size) { push(u32x4{ZC_RLE, size, byte, 0u}); }
  __device__ __forceinline__ bool frame_end(uint32_t frame_start, uint32_t want) {
    push(u32x4{ZC_FRAME, frame_start, want, 0u});
    drain();
    return !aborted && zq_load(&q.verdict) != 0u;
  }
  // false: the page was given up
  __device__ __forceinline__ bool finish() {
    push(u32x4{ZC_DONE, 0u, 0u, 0u});
    drain();
    return !aborted;
  }
};

// The consumer's side: the second wave of the workgroup.
__device__ __forceinline__ void zstd_consume(ZQueue& q, ZExec& x) {
  uint32_t tail = 0, head_seen = 0, pend_lt = 0, pend_rle = 0, pend_pos = 0;
  const PQD_GLOBAL uint8_t* pend_base = nullptr;
  for (;;) {
    uint32_t rounds = 0;
    bool given_up = false;
    while (head_seen == tail) {
      head_seen = zq_load(&q.head);
      if (head_seen == tail) {
        __builtin_amdgcn_s_sleep(2);
        if (++rounds > ZQ_PATIENCE || zq_load(&q.abort)) {
          zq_store(&q.abort, 1u);
          given_up = true;
          break;
        }
      }
    }
    if (given_up) break;
    const u32x4 e = q.e[tail & (ZQ_N - 1u)];
    const uint32_t kind = uni(e.x) & 7u, a = uni(e.y), b = uni(e.z), c = uni(e.w), x0 = uni(e.x);
    ++tail;
    zq_store(&q.tail, tail);
    if (kind == ZC_SEQ) {
      if (a) x.literals(a);
      x.match(c, b);
    } else if (kind == ZC_LITS) {
      x.literals(a);
    } else if (kind == ZC_SETLIT) {
      pend_lt = (x0 >> 3) & 3u;
      pend_rle = (x0 >> 8) & 0xFFu;
      pend_base = (const PQD_GLOBAL uint8_t*)(uintptr_t)(((uint64_t)b << 32) | a);
      pend_pos = c;
    } else if (kind == ZC_SETLIT2) {
      x.set_literals(pend_lt, pend_base, pend_pos, a, pend_rle);
    } else if (kind == ZC_RAW) {
      x.raw(b, a);
    } else if (kind == ZC_RLE) {
      x.rle_block(b, a);
    } else if (kind == ZC_FRAME) {
      const bool fine = x.frame_end(a, b);
      if (x.lane == 0) q.verdict = fine ? 1u : 0u;
    }
    if (kind == ZC_DONE || kind == ZC_FRAME) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    zq_store(&q.done, tail);
    if (kind == ZC_DONE) break;
  }
}

// One page: the serial part.  sh, s_stage: the wave's LDS; sink: where the bytes are moved (above).
template <class SINK>
__device__ __forceinline__ void zstd_page(ZShared& sh, uint8_t* s_stage, const pqd_decompress_desc& d, SINK& sink, uint32_t lane, uint32_t page,
                                          pqd_page_result* __restrict__ results) {
  if (lane < 36u) sh.ll_code[lane] = Z_LL_BASE[lane] | ((uint32_t)Z_LL_BITS[lane] << 24);
  if (lane < 53u) sh.ml_code[lane] = Z_ML_BASE[lane] | ((uint32_t)Z_ML_BITS[lane] << 24);
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
  PQD_GLOBAL uint8_t* dst = (PQD_GLOBAL uint8_t*)d.dst;
  const uint32_t cap = d.dst_len, in_end = in.end;
  uint32_t ip = skew, op = 0;
  uint32_t lit_busy_lo = cap;  // decoded literals that the other wave may not have read yet lie from here to the page's end
  int32_t status = PQD_OK;

  while (ip < in_end && status == PQD_OK) {  // frames one behind the other
    if (in_end - ip < 4u) {
      status = PQD_DECODE;
      break;
    }
    const uint32_t magic = zbyte(in, ip) | (zbyte(in, ip + 1) << 8) | (zbyte(in, ip + 2) << 16) | (zbyte(in, ip + 3) << 24);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {  // skippable frame
      if (in_end - ip < 8u) {
        status = PQD_DECODE;
        break;
      }
      const uint32_t sz = zbyte(in, ip + 4) | (zbyte(in, ip + 5) << 8) | (zbyte(in, ip + 6) << 16) | (zbyte(in, ip + 7) << 24);
      if (in_end - ip - 8u < sz) {
        status = PQD_DECODE;
        break;
      }
      ip += 8u + sz;
      continue;
    }
    if (magic != 0xFD2FB528u || in_end - ip < 5u) {
      status = PQD_DECODE;
      break;
    }
    const uint32_t fhd = zbyte(in, ip + 4);
    uint32_t p = ip + 5;
    const uint32_t single = (fhd >> 5) & 1u, checksum = (fhd >> 2) & 1u, did = fhd & 3u, fcs = fhd >> 6;
    if (fhd & 0x08u) {
      status = PQD_DECODE;
      break;
    }
    const uint32_t did_bytes = did == 3u ? 4u : did, fcs_bytes = fcs == 0u ? single : 1u << fcs;
    if (in_end - p < (single ? 0u : 1u) + did_bytes + fcs_bytes) {
      status = PQD_DECODE;
      break;
    }
    if (!single) ++p;  // window size: no limit is enforced
    uint32_t dict_id = 0;
    for (uint32_t i = 0; i < did_bytes; ++i) dict_id |= zbyte(in, p++) << (8 * i);
    if (dict_id != 0) {  // a dictionary is needed
      status = PQD_DECODE;
      break;
    }
    uint64_t content = 0;
    for (uint32_t i = 0; i < fcs_bytes; ++i) content |= (uint64_t)zbyte(in, p++) << (8 * i);
    if (fcs_bytes == 2) content += 256;
    lz_sync();
    sh.have_ll = sh.have_of = sh.have_ml = sh.huf_valid = 0;
    sh.rep[0] = 1;
    sh.rep[1] = 4;
    sh.rep[2] = 8;
    lz_sync();
    const uint32_t frame_start = op;
    for (;;) {  // blocks
      if (in_end - p < 3u) {
        status = PQD_DECODE;
        break;
      }
      const uint32_t bh = zbyte(in, p) | (zbyte(in, p + 1) << 8) | (zbyte(in, p + 2) << 16);
      p += 3;
      const uint32_t last = bh & 1u, type = (bh >> 1) & 3u, size = bh >> 3;
      if (type == 0) {  // raw
        if (in_end - p < size || size > cap - op) {
          status = PQD_DECODE;
          break;
        }
        sink.raw(p, size);
        op += size;
        p += size;
      } else if (type == 1) {  // one byte, repeated
        if (in_end - p < 1u || size > cap - op) {
          status = PQD_DECODE;
          break;
        }
        sink.rle_block(zbyte(in, p), size);
        op += size;
        p += 1;
      } else if (type == 2) {
        if (in_end - p < size || size > ZS_BLOCK_MAX || size < 1u) {
          status = PQD_DECODE;
          break;
        }
        // ---- literals
        const uint32_t b0 = zbyte(in, p);
        const uint32_t lt = b0 & 3u, sf = (b0 >> 2) & 3u;
        uint32_t hdr, regen, comp = 0, streams = 1;
        if (lt < 2) {
          if (sf == 0 || sf == 2) {
            hdr = 1;
            regen = b0 >> 3;
          } else if (sf == 1) {
            hdr = 2;
            if (size < 2) {
              status = PQD_DECODE;
              break;
            }
            regen = (b0 >> 4) | (zbyte(in, p + 1) << 4);
          } else {
            hdr = 3;
            if (size < 3) {
              status = PQD_DECODE;
              break;
            }
            regen = (b0 >> 4) | (zbyte(in, p + 1) << 4) | (zbyte(in, p + 2) << 12);
          }
        } else {
          const uint32_t bits = sf < 2 ? 10u : sf == 2 ? 14u : 18u;
          hdr = sf < 2 ? 3u : sf == 2 ? 4u : 5u;
          streams = sf == 0 ? 1u : 4u;
          if (size < hdr) {
            status = PQD_DECODE;
            break;
          }
          uint64_t h = 0;
          for (uint32_t i = 0; i < hdr; ++i) h |= (uint64_t)zbyte(in, p + i) << (8 * i);
          regen = (uint32_t)(h >> 4) & ((1u << bits) - 1u);
          comp = (uint32_t)(h >> (4 + bits)) & ((1u << bits) - 1u);
        }
        if (regen > ZS_BLOCK_MAX) {
          status = PQD_DECODE;
          break;
        }
        uint32_t pos = hdr;  // within the block
        // where the literals are read from: the input (raw), one byte (RLE), or the page's end (Huffman)
        const PQD_GLOBAL uint8_t* lit_base = in.base;
        uint32_t lit_end = 0;
        uint32_t lit_pos = 0;  // position of the first literal, counted from lit_base
        uint32_t rle_byte = 0;
        if (lt == 0) {
          if (size - pos < regen) {
            status = PQD_DECODE;
            break;
          }
          lit_end = in.end;
          lit_pos = p + pos;
          pos += regen;
        } else if (lt == 1) {
          if (size - pos < 1u) {
            status = PQD_DECODE;
            break;
          }
          rle_byte = zbyte(in, p + pos);
          pos += 1;
        } else {
          if (size - pos < comp) {
            status = PQD_DECODE;
            break;
          }
          uint32_t q = p + pos, left = comp;
          if (lt == 2) {
            const uint32_t used = zhuf_tree(sh, in, q, left);
            if (used == ~0u) {
              status = PQD_DECODE;
              break;
            }
            q += used;
            left -= used;
          } else if (!sh.huf_valid) {
            status = PQD_DECODE;
            break;
          }
          // the literals are decoded into the last `regen` bytes of the destination (when the page has room for
          // them at all: a block cannot hold more literals than the page has bytes left)
          if (regen > cap - op) {
            status = PQD_DECODE;
            break;
          }
          // (With a second wave moving the bytes, the blocks before this one may still be reading their literals at the
          // page's end: this block's go in front of theirs while a whole block's output still fits in front of that,
          // and the other wave is waited for otherwise.)
          uint32_t lit_hi = cap;
          if (SINK::PIPELINED) {
            if (regen <= lit_busy_lo && (uint64_t)op + ZS_BLOCK_MAX + regen <= lit_busy_lo) {
              lit_hi = lit_busy_lo;
            } else {
              sink.drain();
            }
            lit_busy_lo = lit_hi - regen;
          }
          PQD_GLOBAL uint8_t* lit_out = dst + (lit_hi - regen);
          const PQD_GLOBAL uint8_t* sbase = in.base + q;
          int32_t hst = PQD_OK;
          if (streams == 1) {
            if (lane == 0) sh.lane_status[0] = zhuf_stream(sh, sbase, left, lit_out, regen);
            lz_sync();
            hst = sh.lane_status[0];
          } else {
            if (left < 6u) {
              status = PQD_DECODE;
              break;
            }
            const uint32_t s1 = zbyte(in, q) | (zbyte(in, q + 1) << 8), s2 = zbyte(in, q + 2) | (zbyte(in, q + 3) << 8),
                           s3 = zbyte(in, q + 4) | (zbyte(in, q + 5) << 8);
            const uint32_t each = (regen + 3u) / 4u;
            if (6u + s1 + s2 + s3 > left || each * 3u > regen) {
              status = PQD_DECODE;
              break;
            }
            const uint32_t s4 = left - 6u - s1 - s2 - s3;
            if (lane < 4) {
              const uint32_t off = lane == 0 ? 0u : lane == 1 ? s1 : lane == 2 ? s1 + s2 : s1 + s2 + s3;
              const uint32_t len = lane == 0 ? s1 : lane == 1 ? s2 : lane == 2 ? s3 : s4;
              const uint32_t cnt = lane < 3 ? each : regen - 3u * each;
              sh.lane_status[lane] = zhuf_stream(sh, sbase + 6u + off, len, lit_out + lane * each, cnt);
            }
            lz_sync();
            // the first stream that fails decides (the streams are decoded in order by the format's decoder)
            for (int k = 3; k >= 0; --k) hst = sh.lane_status[k] != PQD_OK ? sh.lane_status[k] : hst;
          }
          if (hst != PQD_OK) {
            status = hst;
            break;
          }
          pos += comp;
          // the decoded literals: stored by lanes of this wave, read back through memory
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          lit_base = (const PQD_GLOBAL uint8_t*)((uintptr_t)lit_out & ~(uintptr_t)15);
          lit_pos = (uint32_t)((uintptr_t)lit_out & 15u);
          lit_end = lit_pos + regen;
        }
        sink.set_literals(lt, lit_base, lit_pos, lit_end, rle_byte);
        uint32_t lit_left = regen;
        // ---- sequences
        if (size - pos < 1u) {
          status = PQD_DECODE;
          break;
        }
        uint32_t n_seq = zbyte(in, p + pos);
        pos += 1;
        if (n_seq >= 128u) {
          if (n_seq == 255u) {
            if (size - pos < 2u) {
              status = PQD_DECODE;
              break;
            }
            n_seq = zbyte(in, p + pos) + (zbyte(in, p + pos + 1) << 8) + 0x7F00u;
            pos += 2;
          } else {
            if (size - pos < 1u) {
              status = PQD_DECODE;
              break;
            }
            n_seq = ((n_seq - 128u) << 8) + zbyte(in, p + pos);
            pos += 1;
          }
        }
        if (n_seq) {
          if (size - pos < 1u) {
            status = PQD_DECODE;
            break;
          }
          const uint32_t modes = zbyte(in, p + pos);
          pos += 1;
          if (modes & 3u) {
            status = PQD_DECODE;
            break;
          }
          uint32_t used = 0;
          for (uint32_t t = 0; t < 3u && used != ~0u; ++t) {  // literal lengths, offsets, match lengths
            u32x2* tab = t == 0 ? sh.ll : t == 1 ? sh.of : sh.ml;
            uint32_t& tlog = t == 0 ? sh.log_ll : t == 1 ? sh.log_of : sh.log_ml;
            uint32_t& have = t == 0 ? sh.have_ll : t == 1 ? sh.have_of : sh.have_ml;
            const PQD_GLOBAL int16_t* def = (const PQD_GLOBAL int16_t*)(t == 0 ? Z_LL_DEFAULT : t == 1 ? Z_OF_DEFAULT : Z_ML_DEFAULT);
            used = zseq_table(sh, in, tab, t, tlog, have, (modes >> (6u - 2u * t)) & 3u, p + pos, size - pos, def,
                              t == 0 ? 36u : t == 1 ? 29u : 53u, t == 1 ? 5u : 6u, t == 0 ? 35u : t == 1 ? 31u : 52u, t == 1 ? 8u : 9u);
            if (used != ~0u) pos += used;
          }
          if (used == ~0u) {
            status = PQD_DECODE;
            break;
          }
          lz_sync();
          const uint32_t sn = size - pos;
          if (sn == 0) {
            status = PQD_DECODE;
            break;
          }
          const uint32_t last_byte = zbyte(in, p + size - 1);
          if (last_byte == 0) {
            status = PQD_DECODE;
            break;
          }
          BackWords bb;
          bb.words = (const PQD_GLOBAL uint32_t*)in.base;
          bb.n_words = (int32_t)((in_end + 3u) >> 2);
          bb.w0 = (int32_t)((p + pos) >> 2);
          bb.lane = lane;
          // bits not yet read, counted from the first bit of word w0
          int32_t bit = (int32_t)(((p + pos) & 3u) * 8u + (sn - 1u) * 8u + zhighbit(last_byte));
          const int32_t bit_end = (int32_t)(((p + pos) & 3u) * 8u);
          bb.start(bit);
          const uint32_t log_ll = uni(sh.log_ll), log_of = uni(sh.log_of), log_ml = uni(sh.log_ml);
          bit -= (int32_t)log_ll;
          uint32_t sl = bb.field(bit, log_ll);
          bit -= (int32_t)log_of;
          uint32_t so = bb.field(bit, log_of);
          bit -= (int32_t)log_ml;
          uint32_t sm = bb.field(bit, log_ml);
          if (bit < bit_end) {
            status = PQD_DECODE;
            break;
          }
          uint32_t rep0 = uni(sh.rep[0]), rep1 = uni(sh.rep[1]), rep2 = uni(sh.rep[2]);
          u32x2 eo = sh.of[so], em = sh.ml[sm], el = sh.ll[sl];
#ifdef ZS_PROF
          uint64_t zp[3] = {0, 0, 0}, zt = __builtin_readcyclecounter(), zfar_cycles = 0;
          uint32_t zfar[4] = {0, 0, 0, 0};
#define ZS_STAMP(k) { const uint64_t now_ = __builtin_readcyclecounter(); zp[k] += now_ - zt; zt = now_; }
#else
#define ZS_STAMP(k)
#endif
          for (uint32_t i = 0; i < n_seq; ++i) {
            bb.need(bit);
            // where the six fields lie follows from the three entries alone
            const uint32_t oy = uni(eo.y), my = uni(em.y), ly = uni(el.y), m_base = uni(em.x), l_base = uni(el.x);
            const uint32_t ob = oy & 0xFFu, mb = my & 0xFFu, lb = ly & 0xFFu;
            const int32_t at_o = bit - (int32_t)ob, at_l = at_o - (int32_t)(mb + lb);
            const uint32_t ov = (1u << ob) + bb.field(at_o, ob);
            // the bits behind the two length baselines lie side by side (32 at most): one look at the stream for both
            const uint32_t both = bb.raw(at_l);
            const uint32_t ml = m_base + ((both >> lb) & ((1u << mb) - 1u));
            const uint32_t ll = l_base + (both & ((1u << lb) - 1u));
            bit = at_l;
            if (i + 1 < n_seq) {
              // ... and so do the bits of the three next states (26 at most)
              const uint32_t nl = (ly >> 8) & 0xFFu, nm = (my >> 8) & 0xFFu, no = (oy >> 8) & 0xFFu;
              bit = at_l - (int32_t)(nl + nm + no);
              const uint32_t three = bb.raw(bit);
              sl = (ly >> 16) + ((three >> (nm + no)) & ((1u << nl) - 1u));
              sm = (my >> 16) + ((three >> no) & ((1u << nm) - 1u));
              so = (oy >> 16) + (three & ((1u << no) - 1u));
              // (the next sequence's entries are on their way while this one's bytes are moved)
              eo = sh.of[so];
              em = sh.ml[sm];
              el = sh.ll[sl];
            }
            if (bit < bit_end) {
              status = PQD_DECODE;
              break;
            }
            // the offset: a new one (the three kept ones move down), or one of the kept ones (RFC 8878 3.1.1.5), without branches
            const bool fresh = ov > 3u;
            const uint32_t idx = ov - 1u + (ll == 0 ? 1u : 0u);  // (meaningful when not fresh: 0..3)
            const uint32_t kept = idx == 0u ? rep0 : idx == 1u ? rep1 : idx == 2u ? rep2 : rep0 - 1u;
            const uint32_t offset = fresh ? ov - 3u : kept;
            rep2 = fresh || idx > 1u ? rep1 : rep2;
            rep1 = fresh || idx > 0u ? rep0 : rep1;
            rep0 = offset;
            if (ll > lit_left || ll > cap - op) {
              status = PQD_DECODE;
              break;
            }
            ZS_STAMP(0)
            // matches reach back into the frame only
            if (offset == 0u || offset > op + ll - frame_start || ml > cap - op - ll) {
              sink.literals(ll);  // (the literals in front of the match that fails are delivered)
              op += ll;
              status = PQD_DECODE;
              break;
            }
#ifdef ZS_PROF
            zfar[0] += offset > ZS_RING ? 1u : 0u;
            zfar[1] += offset > 4u * ZS_RING ? 1u : 0u;
            zfar[2] += offset < 64u ? 1u : 0u;
            zfar[3] += ll == 0u ? 1u : 0u;
#endif
            sink.sequence(ll, offset, ml);
            lit_left -= ll;
            op += ll;
            op += ml;
            ZS_STAMP(2)
          }
#ifdef ZS_PROF
          if (page == 0 && lane == 0) printf("block: %u sequences, cycles per sequence: decode %llu, literals %llu, match %llu; offsets beyond the ring %u (%llu cycles each), beyond four rings %u, below 64: %u, without literals %u\n", n_seq, zp[0] / n_seq, zp[1] / n_seq, zp[2] / n_seq, zfar[0], zfar[0] ? zfar_cycles / zfar[0] : 0ull, zfar[1], zfar[2], zfar[3]);
#endif
          lz_sync();
          sh.rep[0] = rep0;
          sh.rep[1] = rep1;
          sh.rep[2] = rep2;
          lz_sync();
          if (status != PQD_OK) break;
          if (bit != bit_end) {
            status = PQD_DECODE;
            break;
          }
        } else if (pos != size) {
          status = PQD_DECODE;
          break;
        }
        if (lit_left > cap - op) {
          status = PQD_DECODE;
          break;
        }
        sink.literals(lit_left);
        op += lit_left;
        p += size;
      } else {
        status = PQD_DECODE;
        break;
      }
      if (last) break;
    }
    if (status != PQD_OK) break;
    if (fcs_bytes != 0 && content != (uint64_t)(op - frame_start)) {
      status = PQD_DECODE;
      break;
    }
    if (checksum) {
      if (in_end - p < 4u) {
        status = PQD_DECODE;
        break;
      }
      const uint32_t want = zbyte(in, p) | (zbyte(in, p + 1) << 8) | (zbyte(in, p + 2) << 16) | (zbyte(in, p + 3) << 24);
      if (!sink.frame_end(frame_start, want)) {
        status = PQD_DECODE;
        break;
      }
      p += 4;
    }
    ip = p;
  }
  if (!sink.finish()) status = PQD_DECODE;
  if (status == PQD_OK && op != cap) status = PQD_DECODE;  // the page header announced cap bytes
  if (lane == 0) {
    pqd_page_result r = {status, op, ip - skew, 0u};
    results[page] = r;
  }
}

// PIPE true: two waves per page, the first decodes, the second moves the bytes.  PIPE false: one wave per page does
// everything.
// RING: bytes of output kept in LDS for the matches (farther ones are read back from memory behind a wait for the
// wave's stores): 8 KiB where seven pages share a CU, more where fewer do.
template <bool PIPE, uint32_t RING>
__global__ __launch_bounds__(PIPE ? 128 : 64) void k_zstd_pages(const pqd_decompress_desc* __restrict__ descs,
                                                                pqd_page_result* __restrict__ results, uint32_t n_pages) {
  __shared__ ZShared sh;
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[RING];
  __shared__ __attribute__((aligned(16))) uint8_t s_stage[LZ_STAGE];
  __shared__ __attribute__((aligned(16))) uint8_t s_lits[LZ_STAGE];
  __shared__ ZQueue queue;
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_decompress_desc d = descs[page];
  if (d.codec != (uint32_t)PQD_CODEC_ZSTD) return;  // another kernel's page
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (d.src_len > 0x7fffffffu) {
    if (threadIdx.x == 0) {
      pqd_page_result r = {PQD_BAD_ARG, 0u, 0u, 0u};
      results[page] = r;
    }
    return;
  }
  ZExec exec;
  exec.init(s_ring, RING, s_lits, d, lane);
  if constexpr (PIPE) {
    if (threadIdx.x == 0) queue.head = queue.tail = queue.done = queue.verdict = queue.abort = 0u;
    __syncthreads();  // (the one barrier of the workgroup: the two waves go separate ways behind it)
    if (wave == 1u) {
      zstd_consume(queue, exec);
    } else {
      ZPush sink = {queue, 0u, 0u, false};
      zstd_page(sh, s_stage, d, sink, lane, page, results);
    }
  } else {
    ZDirect sink = {exec};
    zstd_page(sh, s_stage, d, sink, lane, page, results);
  }
}

// Two waves per page; the fewer the pages of the call, the more of a page's output is kept in LDS (63 / 47 / 39 KB of
// LDS a page: two, three, four pages a CU).  One wave per page does no better at any size of call (1792 pages of 1 MiB:
// 138 / 200 / 202 ms on text / small integers / doubles against 139 / 173 / 179 with two): it is kept for comparison.
constexpr uint32_t ZS_BIG_RING_MAX_PAGES = 512, ZS_MID_RING_MAX_PAGES = 768;

hipError_t launch_zstd_pages(const pqd_decompress_desc* descs, pqd_page_result* results, uint32_t n_pages, hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  const char* env = getenv("PQD_ZSTD_WAVES");  // 1: one wave per page (testing)
  const bool pipe = !(env && atoi(env) == 1);
  if (!pipe)
    hipLaunchKernelGGL((k_zstd_pages<false, ZS_RING>), dim3(n_pages), dim3(64), 0, stream, descs, results, n_pages);
  else if (n_pages <= ZS_BIG_RING_MAX_PAGES)
    hipLaunchKernelGGL((k_zstd_pages<true, 32768u>), dim3(n_pages), dim3(128), 0, stream, descs, results, n_pages);
  else if (n_pages <= ZS_MID_RING_MAX_PAGES)
    hipLaunchKernelGGL((k_zstd_pages<true, 16384u>), dim3(n_pages), dim3(128), 0, stream, descs, results, n_pages);
  else
    hipLaunchKernelGGL((k_zstd_pages<true, ZS_RING>), dim3(n_pages), dim3(128), 0, stream, descs, results, n_pages);
  return hipGetLastError();
}

}  // namespace pqd
