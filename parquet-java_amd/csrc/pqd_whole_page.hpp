// Run-heavy dictionary pages decoded whole: ids and dictionary lookups in one pass, by one workgroup.
//
// value = dictionary.decodeToX(id) for the ids of an RLE / bit-packing hybrid stream (DictionaryValuesReader.java:75-118
// over RunLengthBitPackingHybridDecoder.readNext, parquet-column/.../values/rle/RunLengthBitPackingHybridDecoder.java:80-109).
// A page of sorted or clustered ids is a few hundred RLE runs with a bit-packed group of eight where two runs meet: two
// or three kilobytes of stream for 20000 values.  Cut into chunks of 231 segments such a page becomes five or six small
// chunks, each of which pays the fixed costs of one (a sweep of the dictionary, or a trip through the chunk pool), and one
// wave of the plan kernel walks its 1200 runs.  Here the whole stream sits in LDS and is parsed the way k_rle_small_pages
// parses level pages: every byte position is tried as a run header, lane exchanges resolve where a chain that enters a
// 64-byte block leaves it, one thread hops over the blocks, every thread lists the runs of its block; a run table and a
// bitmap of run starts then give every value its run by a popcount.  The dictionary is read from memory (L2): the ids of
// such a page are few and close together.
// Covered: streams that are at most half of what the page's values would take bit-packed (taken through LDS a window of
// 8 KiB and 32768 values at a time: a window ends in front of the first run that does not fit, or inside an RLE run,
// whose remaining values are then stored without a window), id widths 1..24, run headers of up to three bytes, bit-packed
// runs shorter than a window.  Small pages (32 KiB of stream, 32768 values) of dictionary ids are sent here by the plan
// kernel because a plan costs them more; long pages of many short runs (a million values in 25000 runs) because one wave
// planning them run by run is the slowest part of their launch.  Anything else, and every malformed stream (zero-length
// runs, runs cut off by the end of the stream, a stream that ends short of the page's values), returns false: the
// caller decodes the page (again, where windows had been written) with the kernel that knows every rule of the reference.
// Traffic: stream bytes in, ELEM bytes out per value; the dictionary entries come from L2.
#pragma once

#include "pqd_device.hpp"
#include "pqd_rle_plan.hpp"
#include "../../include/pqd.h"

namespace pqd {

constexpr int WP_THREADS = 256;
constexpr uint32_t WP_BLOCK = 64;  // stream bytes per block (one lane per byte in the candidate step)
constexpr uint32_t WP_MAX_BYTES = 8192;     // stream bytes in LDS at a time (a window)
constexpr uint32_t WP_MAX_STREAM = 32768;   // stream bytes of a page: up to four windows, one after the other
constexpr uint32_t WP_BLOCKS = WP_MAX_BYTES / WP_BLOCK;
constexpr uint32_t WP_MAX_VALUES = 32768;
constexpr uint32_t WP_MAX_WIDTH = 24;
constexpr uint32_t WP_COMPLEX = 0xFFFFu, WP_END = 0xFFFEu, WP_STOP = 0xFFFDu;  // (above every stream position of a window)

// The two shapes of the routine: a workgroup of 256 threads with windows of 8 KiB and 32768 values (five of them fit a
// CU: thousands of small pages), and one of 1024 threads with windows of 16 KiB and 262144 values that has a CU to
// itself (a launch of few long pages: sixteen waves hide each other's LDS round trips, and the chain through a window
// is followed 256 bytes at a hop).
struct WpSmall {
  static constexpr int THREADS = WP_THREADS;
  static constexpr uint32_t BYTES = WP_MAX_BYTES, VALUES = WP_MAX_VALUES, VAL_RUNS = 1360, BIAS = 1u << 20;  // BIAS: above first value * w
  static constexpr bool SUPER = false;
};
struct WpLong {
  static constexpr int THREADS = 1024;
  static constexpr uint32_t BYTES = 16384, VALUES = 262144, VAL_RUNS = 2728, BIAS = 1u << 23;
  static constexpr bool SUPER = true;
};
constexpr uint32_t WP_SUPER = 256;  // bytes per hop of the chain where SUPER is set

template <class C>
struct alignas(16) WholePageLdsT {
  uint8_t pg[C::BYTES + 96];    // the stream, zeros behind it
  uint16_t tab[C::BYTES + 64];  // where a chain entering at each byte leaves its block; later the run table (as words)
  uint16_t tab2[C::SUPER ? C::BYTES + 64 : 8];   // ... leaves its 256 bytes
  uint16_t tab16[C::SUPER ? C::BYTES + 64 : 8];  // ... leaves its 16 bytes
  uint32_t flags[C::VALUES / 32 + 4];  // bit v: a run starts at value v
  uint16_t wbase[C::VALUES / 64 + 4];  // run starts before each 64-bit word of the bitmap
  uint16_t entry_of[C::BYTES / WP_BLOCK];  // where the chain enters each block
  uint16_t sentry[C::BYTES / WP_SUPER + 1];  // ... each 256 bytes
  uint16_t entry16[C::SUPER ? C::BYTES / 16 : 8];  // ... each 16 bytes
  uint32_t wave_totals[C::THREADS / 64];
  uint32_t ok, consumed, oob_at, rle_oob, n_eff;
  uint32_t cut_at, cut_nv, carry_n, carry_val;  // a window that ends at its last value: where the next one starts
  static_assert(C::BYTES / WP_BLOCK <= (uint32_t)C::THREADS, "a thread per block");
  static_assert(!C::SUPER || C::BYTES / 16 <= (uint32_t)C::THREADS, "a thread per 16 bytes");
  static_assert((C::BYTES / 2) * 4 <= sizeof(uint16_t) * (C::BYTES + 64), "the run table fits where the exits were");
  static_assert(C::VAL_RUNS * 12u <= sizeof(uint16_t) * (C::BYTES + 64) && (C::VAL_RUNS * 4u) % 8u == 0, "table and values fit where the exits were");
  static_assert((uint64_t)C::VALUES * 24u < C::BIAS && C::BIAS + C::BYTES * 8u < (1u << 31), "an entry holds the biased bit position");
  static_assert(C::BYTES + 64 < 0xFFFDu, "positions stay below the marks");
};
using WholePageLds = WholePageLdsT<WpSmall>;

// Pages the plan kernel leaves to dict_whole_page (max_bytes: the launch's limit, at most WP_MAX_STREAM; 0: none).
__host__ __device__ inline bool whole_page_candidate(uint32_t stream_bytes, uint32_t n_values, uint32_t w, uint32_t dict_len,
                                                     uint32_t max_bytes) {
  return w >= 1u && w <= WP_MAX_WIDTH && n_values != 0u && n_values <= WP_MAX_VALUES && stream_bytes != 0u &&
         stream_bytes <= max_bytes && stream_bytes <= WP_MAX_STREAM && dict_len != 0u &&
         (uint64_t)stream_bytes * 16u <= (uint64_t)n_values * w;
}
// Long run-heavy pages: more values or stream bytes than the rule above takes, and runs of fewer than 64 values on
// average (pages of longer runs are planned quickly, and their expansion is the faster one).
constexpr uint32_t WP_LONG_MAX_VALUES = 1u << 30;
__host__ __device__ inline bool whole_page_long_candidate(uint32_t stream_bytes, uint32_t n_values, uint32_t w, uint32_t dict_len,
                                                          uint32_t max_bytes) {
  return max_bytes != 0u && w >= 1u && w <= WP_MAX_WIDTH && n_values <= WP_LONG_MAX_VALUES && dict_len != 0u &&
         (n_values > WP_MAX_VALUES || stream_bytes > WP_MAX_STREAM) && (uint64_t)stream_bytes * 16u <= (uint64_t)n_values * w &&
         (uint64_t)stream_bytes * 32u >= (uint64_t)n_values;
}

// The stream of a page and what stands in front of it (RunLengthBitPackingHybridValuesReader.java:41-45: four bytes of
// length; DictionaryValuesReader.java:49-63: a byte of width; levels of V2 pages and booleans: nothing).
struct WholeStream {
  const PQD_GLOBAL uint8_t* src;
  uint32_t n, w, prefix;
  uint32_t consumed_all;  // != ~0u: what the page reports as consumed whatever the runs take (the length prefix says it)
  bool ok;
};
__device__ __forceinline__ WholeStream whole_stream_of(const pqd_page_desc& d, bool has_dict) {
  WholeStream s = {(const PQD_GLOBAL uint8_t*)d.data, d.data_len, d.bit_width, 0u, ~0u, d.data != nullptr};
  if (has_dict || (d.flags & PQD_F_WIDTH_PREFIX)) {
    s.ok = s.ok && d.data_len >= 2u;
    if (s.ok) s.w = s.src[0];
    s.prefix = 1u;
  } else if (d.flags & PQD_F_LENGTH_PREFIX) {
    s.ok = s.ok && d.data_len >= 4u;
    uint32_t len = 0;
    if (s.ok) len = (uint32_t)s.src[0] | ((uint32_t)s.src[1] << 8) | ((uint32_t)s.src[2] << 16) | ((uint32_t)s.src[3] << 24);
    s.ok = s.ok && (int32_t)len >= 0 && len <= d.data_len - 4u;
    s.prefix = 4u;
    s.n = 4u + len;
    s.consumed_all = 4u + len;
  }
  s.ok = s.ok && s.n > s.prefix;
  s.src += s.prefix;
  s.n = s.ok ? s.n - s.prefix : 0u;
  return s;
}

struct WholeRun {
  uint32_t len, nvals, nb;
  bool packed, complex;
};

// The run whose header starts with the bytes `lo`; vb = bytes of an RLE run's value.  Headers of up to three bytes:
// counts below 2^20, more than a page of 32768 values needs.
__device__ __forceinline__ WholeRun whole_run(uint32_t lo, uint32_t w, uint32_t vb) {
  WholeRun r;
  const uint32_t c0 = (lo >> 7) & 1u, c01 = c0 & (lo >> 15), c012 = c01 & (lo >> 23);
  const uint32_t h = (lo & 0x7Fu) | (c0 ? (lo >> 1) & 0x3F80u : 0u) | (c01 ? (lo >> 2) & 0x1FC000u : 0u);
  const uint32_t cnt = h >> 1;
  r.nb = 1u + c0 + c01;
  r.packed = (h & 1u) != 0;
  r.complex = c012 != 0 || cnt == 0;
  r.len = r.nb + (r.packed ? cnt * w : vb);
  r.nvals = r.packed ? cnt * 8u : cnt;
  return r;
}

// Exclusive prefix of `v` over the workgroup's threads and the total.
template <int THREADS>
__device__ __forceinline__ uint32_t wp_exclusive(uint32_t v, uint32_t* wave_totals, uint32_t& total) {
  const uint32_t inc = wave_scan_inclusive(v);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  __syncthreads();
  if (lane == 63) wave_totals[wave] = inc;
  __syncthreads();
  uint32_t before = 0;
  total = 0;
#pragma unroll
  for (uint32_t k = 0; k < (uint32_t)THREADS / 64u; ++k) {
    const uint32_t t = wave_totals[k];
    before += k < wave ? t : 0u;
    total += t;
  }
  return before + inc - v;
}

// (Where the looked-up values of the RLE runs sit when a window has at most VAL_RUNS runs: behind that many table
// entries, in the space the block exits used.)

// The dictionary entries of values v0 .. v0 + 3 of the window (v0 a multiple of four below n_values, the window's number of
// values; v_base: the page's values before the window); x[k].y is
// meaningless for 4-byte dictionaries.  FAST: the entries of RLE runs were looked up once per run and are read from
// LDS; only values of bit-packed groups are fetched from the dictionary.
template <int ELEM, bool FAST, bool IDENT, class C>
__device__ __forceinline__ void wp_unit(WholePageLdsT<C>& L, const pqd_page_desc& d, uint32_t v0, uint32_t n_values, uint32_t v_base,
                                        uint32_t w, uint32_t mask, u32x2 x[4]) {
  const uint32_t* pgw = reinterpret_cast<const uint32_t*>(L.pg);
  const uint32_t* runs = reinterpret_cast<const uint32_t*>(L.tab);
  const uint32_t wd = v0 >> 6, sh = v0 & 63u;  // the four values share a word of the bitmap
  const uint64_t bits = (uint64_t)L.flags[2 * wd] | ((uint64_t)L.flags[2 * wd + 1] << 32);
  uint32_t ri = L.wbase[wd] + (uint32_t)__builtin_popcountll(bits & ((2ull << sh) - 1ull)) - 1u;
  const uint32_t near = (uint32_t)(bits >> sh);  // bits 1..3: a run starts at value v0 + 1..3
  uint32_t vw = v0 * w - C::BIAS;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k) ri += (near >> k) & 1u;
    const uint32_t e = runs[ri];
    if (FAST && !(e >> 31)) {
      if (ELEM == 4)
        x[k].x = (runs + C::VAL_RUNS)[ri];
      else
        x[k] = reinterpret_cast<const u32x2*>(runs + C::VAL_RUNS)[ri];
    } else {
      const uint32_t bit = (e & 0x7FFFFFFFu) + vw;
      const uint32_t i0 = (bit >> 5) & (C::BYTES / 4 - 1u);  // (an RLE entry gives a meaningless but harmless index)
      uint32_t id = (e >> 31) ? funnel_shr(pgw[i0], pgw[i0 + 1], bit & 31u) & mask : e;
      if (IDENT) {
        x[k].x = id;  // readValueDictionaryId: the id is the value
      } else {
        if (id >= d.dict_len) {  // ColumnReaderBase.java:582-616: the first such value ends what the page delivers
          if (v0 + k < n_values) atomicMin(&L.oob_at, v_base + v0 + k);
          id = 0u;
        }
        if (ELEM == 4)
          x[k].x = ((const PQD_GLOBAL uint32_t*)d.dict)[id];
        else
          x[k] = ((const PQD_GLOBAL u32x2_a4*)d.dict)[id];
      }
    }
    vw += w;
  }
}

// Step 5 of dict_whole_page: per step two units of four values per thread, their lookups requested together (n_values:
// the window's values; v_base: the page's values before it).
template <int ELEM, bool FAST, bool IDENT, class C>
__device__ __forceinline__ void wp_values(WholePageLdsT<C>& L, const pqd_page_desc& d, uint32_t n_values, uint32_t v_base, uint32_t w) {
  const uint32_t mask = (1u << w) - 1u;
  constexpr uint32_t STEP = (uint32_t)C::THREADS * 4u;
  for (uint32_t v0 = threadIdx.x * 4u; v0 < n_values; v0 += 2u * STEP) {
    u32x2 x[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t vv = v0 + (uint32_t)u * STEP;
      if (vv < n_values) wp_unit<ELEM, FAST, IDENT, C>(L, d, vv, n_values, v_base, w, mask, x[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t vv = v0 + (uint32_t)u * STEP;
      if (vv >= n_values) continue;
      if (ELEM == 4) {
        PQD_GLOBAL uint32_t* o = (PQD_GLOBAL uint32_t*)d.out + v_base + vv;
        if (vv + 4u <= n_values) {
          const u32x4_a4 q = {x[u][0].x, x[u][1].x, x[u][2].x, x[u][3].x};
          PQD_STREAM_STORE((PQD_GLOBAL u32x4_a4*)o, q);
        } else {
#pragma unroll
          for (uint32_t k = 0; k < 4u; ++k)
            if (vv + k < n_values) o[k] = x[u][k].x;
        }
      } else {
        PQD_GLOBAL u32x2_a4* o = (PQD_GLOBAL u32x2_a4*)d.out + v_base + vv;
        if (vv + 4u <= n_values) {
          const u32x4_a4 q0 = {x[u][0].x, x[u][0].y, x[u][1].x, x[u][1].y}, q1 = {x[u][2].x, x[u][2].y, x[u][3].x, x[u][3].y};
          PQD_STREAM_STORE((PQD_GLOBAL u32x4_a4*)o, q0);
          PQD_STREAM_STORE((PQD_GLOBAL u32x4_a4*)(o + 2), q1);
        } else {
#pragma unroll
          for (uint32_t k = 0; k < 4u; ++k)
            if (vv + k < n_values) {
              const u32x2_a4 e = {x[u][k].x, x[u][k].y};
              o[k] = e;
            }
        }
      }
    }
  }
}

// Step 5 for plain values in the long shape, where most values lie in RLE runs: first every value of an RLE run (a unit
// of four values reads its bitmap word and the runs' entries, nothing of the stream), then the bit-packed runs, a thread
// per run (their first values are kept beside the run table, in tab2).  A value is stored once: units that touch a
// bit-packed run leave its values out.
template <class C>
__device__ __forceinline__ void wp_values_split(WholePageLdsT<C>& L, const pqd_page_desc& d, uint32_t n_values, uint32_t v_base, uint32_t w,
                                                uint32_t n_runs, uint32_t vt, uint32_t n_vt) {
  const uint32_t* runs = reinterpret_cast<const uint32_t*>(L.tab);
  const uint32_t* valof = reinterpret_cast<const uint32_t*>(L.tab2);
  const uint32_t* pgw = reinterpret_cast<const uint32_t*>(L.pg);
  PQD_GLOBAL uint32_t* out = (PQD_GLOBAL uint32_t*)d.out + v_base;
  const uint32_t STEP = n_vt * 4u;
  for (uint32_t v0 = vt * 4u; v0 < n_values; v0 += 2u * STEP) {
    uint32_t e[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t vv = v0 + (uint32_t)u * STEP;
      const uint32_t vc = vv < n_values ? vv : 0u;
      const uint32_t wd = vc >> 6, sh = vc & 63u;  // the four values share a word of the bitmap
      const uint64_t bits = (uint64_t)L.flags[2 * wd] | ((uint64_t)L.flags[2 * wd + 1] << 32);
      uint32_t ri = L.wbase[wd] + (uint32_t)__builtin_popcountll(bits & ((2ull << sh) - 1ull)) - 1u;
      const uint32_t near = (uint32_t)(bits >> sh);  // bits 1..3: a run starts at value vv + 1..3
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k) ri += (near >> k) & 1u;
        e[u][k] = runs[ri];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t vv = v0 + (uint32_t)u * STEP;
      if (vv >= n_values) continue;
      const uint32_t any_packed = (e[u][0] | e[u][1] | e[u][2] | e[u][3]) >> 31;
      if (!any_packed && vv + 4u <= n_values) {
        const u32x4_a4 q = {e[u][0], e[u][1], e[u][2], e[u][3]};
        PQD_STREAM_STORE((PQD_GLOBAL u32x4_a4*)(out + vv), q);
      } else {
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
          if (vv + k < n_values && !(e[u][k] >> 31)) out[vv + k] = e[u][k];
      }
    }
  }
  const uint32_t mask = (1u << w) - 1u;
  for (uint32_t r = vt; r < n_runs; r += n_vt) {
    const uint32_t en = runs[r];
    if (!(en >> 31)) continue;
    const uint32_t a = valof[r], b = valof[r + 1u];
    uint32_t bit = (en & 0x7FFFFFFFu) + a * w - C::BIAS;
    for (uint32_t v = a; v < b; ++v) {
      const uint32_t i0 = (bit >> 5) & (C::BYTES / 4 - 1u);
      out[v] = funnel_shr(pgw[i0], pgw[i0 + 1], bit & 31u) & mask;
      bit += w;
    }
  }
}

// One page by the calling workgroup of WP_THREADS threads; L: its LDS, free for this call.  The stream is taken a window
// of WP_MAX_BYTES at a time; a window ends in front of the first run that does not fit into it, or at its 32768th value.
// false (for all threads): not decoded (what was written is written again by the caller's decoder).
// IDENT: a page of plain hybrid values (levels, booleans, dictionary ids read as ints through
// DictionaryValuesReader.readValueDictionaryId): the same stream, the id itself is the value and no dictionary is looked at.
#ifdef WP_DEBUG
#define WP_FAIL(code)                                                                          \
  do {                                                                                         \
    if (threadIdx.x == 0) printf("whole page: gives up at %d (page of %u values)\n", code, d.num_values); \
    return false;                                                                              \
  } while (0)
#else
#define WP_FAIL(code) return false
#endif
// part / parts: the workgroup is one of `parts` that decode the page together (plain values in the long shape only): each
// parses every window and stores its share of the values; part 0 writes the result.
template <int ELEM, bool IDENT = false, class C = WpSmall>
__device__ bool dict_whole_page(const pqd_page_desc& d, pqd_page_result* result, WholePageLdsT<C>& L, uint32_t part = 0, uint32_t parts = 1) {
  const uint32_t tid = threadIdx.x;
  const uint32_t vt = part * (uint32_t)C::THREADS + tid, n_vt = parts * (uint32_t)C::THREADS;  // this thread among those of all parts
  const uint32_t n_page_values = d.num_values;
  if (!IDENT && d.dict == nullptr) WP_FAIL(1);
  const WholeStream S = whole_stream_of(d, !IDENT);
  if (!S.ok || n_page_values == 0u) WP_FAIL(2);
  const uint32_t n_stream = S.n, w = S.w;
  // (the small shape takes the pages of whole_page_candidate only; the long one anything the rules of either allow)
  if (!whole_page_candidate(n_stream, n_page_values, w, IDENT ? 1u : d.dict_len, WP_MAX_STREAM) &&
      !(C::SUPER && whole_page_long_candidate(n_stream, n_page_values, w, IDENT ? 1u : d.dict_len, WP_MAX_STREAM)))
    WP_FAIL(3);
  if (tid == 0) {
    L.consumed = 0u;
    L.oob_at = ~0u;
  }
#ifdef PQD_PROF
  uint64_t stamp[16];
  int n_stamp = 0;
#define WP_STAMP() \
  if (base == 0u) stamp[n_stamp++] = __builtin_readcyclecounter()
#else
#define WP_STAMP()
#endif
  uint32_t base = 0;    // stream bytes before the window
  uint32_t v_base = 0;  // values before the window
  uint32_t carry_n = 0, carry_val = 0;  // values left of an RLE run whose bytes lie before the window
  for (;;) {
    // ---- 0. what is left of an RLE run that the window before this one ended in: stored as it is
    if (carry_n) {
      const uint32_t take = carry_n < n_page_values - v_base ? carry_n : n_page_values - v_base;
      u32x2 x = {carry_val, 0u};
      if (!IDENT && carry_val < d.dict_len) {  // (beyond the dictionary: the run's first value has reported it)
        if (ELEM == 4)
          x.x = ((const PQD_GLOBAL uint32_t*)d.dict)[carry_val];
        else
          x = ((const PQD_GLOBAL u32x2_a4*)d.dict)[carry_val];
      }
      if (ELEM == 4) {
        PQD_GLOBAL uint32_t* o = (PQD_GLOBAL uint32_t*)d.out + v_base;
        for (uint32_t i = vt; i < take; i += n_vt) o[i] = x.x;
      } else {
        PQD_GLOBAL u32x2_a4* o = (PQD_GLOBAL u32x2_a4*)d.out + v_base;
        const u32x2_a4 e = {x.x, x.y};
        for (uint32_t i = vt; i < take; i += n_vt) o[i] = e;
      }
      v_base += take;
      carry_n = 0;
      if (v_base >= n_page_values) break;
    }
    const bool last = n_stream - base <= C::BYTES;
    const uint32_t N = last ? n_stream - base : C::BYTES;
    const uint32_t n_values = n_page_values - v_base;  // values still wanted
    const uint32_t n_window = n_values < C::VALUES ? n_values : C::VALUES;  // values this window delivers at most
    const PQD_GLOBAL uint8_t* src = S.src + base;
    WP_STAMP();
    // ---- 1. the window -> LDS (zeros behind it).  Whole aligned dwords are read (none lies wholly outside the page) and
    // shifted into place: a piece of 16 stream bytes is five independent loads, whatever the page's address.
    {
      const uintptr_t a = (uintptr_t)src;
      const PQD_GLOBAL uint32_t* wsrc = (const PQD_GLOBAL uint32_t*)(a & ~(uintptr_t)3);
      const uint32_t off = (uint32_t)(a & 3u), sh = off * 8u;
      const uint32_t full = (off + N) >> 2;  // dwords wholly inside [aligned start, end of the window)
      const uint32_t tail = (off + N) & 3u;  // window bytes in dword `full`
      for (uint32_t i = tid * 4u; i * 4u < N + 16u; i += (uint32_t)C::THREADS * 4u) {
        uint32_t x[5];
#pragma unroll
        for (uint32_t k = 0; k < 5u; ++k) {
          x[k] = 0u;
          if (i + k < full)
            x[k] = wsrc[i + k];
          else if (i + k == full && tail)
            x[k] = wsrc[i + k] & ((1u << (8u * tail)) - 1u);
        }
        const u32x4 v = {funnel_shr(x[0], x[1], sh), funnel_shr(x[1], x[2], sh), funnel_shr(x[2], x[3], sh), funnel_shr(x[3], x[4], sh)};
        *reinterpret_cast<u32x4*>(L.pg + i * 4u) = v;
      }
    }
    for (uint32_t i = tid; i < (n_window >> 5) + 4u; i += (uint32_t)C::THREADS) L.flags[i] = 0u;
    if (tid < (C::BYTES / WP_BLOCK)) L.entry_of[tid] = 0xFFFFu;
    if constexpr (C::SUPER) {
      if (tid < C::BYTES / 16u) L.entry16[tid] = 0xFFFFu;
    }
    if (tid == 0) {
      L.ok = 1u;
      L.rle_oob = 0u;
      L.n_eff = N;
      L.cut_at = ~0u;
    }
    __syncthreads();
    const uint32_t vb = (w + 7u) >> 3;
    const uint32_t* pgw = reinterpret_cast<const uint32_t*>(L.pg);

    WP_STAMP();
    // ---- 2. candidates and block exits: a wave takes a 64-byte block at a time, one lane per byte.  Where a chain that
    // enters at a byte leaves the block: five rounds of "continue from where I land" between the lanes (a run is at
    // least two bytes, so 32 hops leave the block from anywhere).  A run that reaches past the window ends the chain:
    // it is malformed in the page's last window (WP_COMPLEX), the first run of the next window otherwise (WP_STOP).
    {
      const uint32_t lane = tid & 63u, wave = tid >> 6;
      const uint32_t n_blocks = (N + WP_BLOCK) / WP_BLOCK;  // the block that holds position N is included
      constexpr int SIDE = 4;  // blocks worked on side by side (their chains of lane exchanges are independent)
      for (uint32_t b0 = wave * SIDE; b0 < n_blocks; b0 += SIDE * ((uint32_t)C::THREADS / 64)) {
        uint32_t to[SIDE];
#pragma unroll
        for (int q = 0; q < SIDE; ++q) {
          const uint32_t p = (b0 + q) * WP_BLOCK + lane;  // blocks past n_blocks: positions past N, complex, never stored
          const uint32_t pw = p < N + 4u ? p >> 2 : 0u;   // (positions past the window read its start: they are never stored)
          const uint32_t lo = funnel_shr(pgw[pw], pgw[pw + 1], (p & 3u) * 8u);
          const WholeRun r = whole_run(lo, w, vb);
          // where the run leads, counted from the block's first byte; the three marks are above every position
          to[q] = r.complex ? WP_COMPLEX : (p + r.len > N ? (last ? WP_COMPLEX : WP_STOP) : lane + r.len);
          if (p == N) to[q] = WP_END;
        }
        // (the long shape: the first three rounds stay inside the lane's 16 bytes, which at most eight runs leave; what they
        // give is kept as well; two more rounds then cross the block's four quarters)
#pragma unroll
        for (int round = 0; round < 5; ++round) {
          uint32_t further[SIDE];
#pragma unroll
          for (int q = 0; q < SIDE; ++q) further[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((to[q] & 63u) << 2), (int)to[q]);
#pragma unroll
          for (int q = 0; q < SIDE; ++q) {
            const bool inside = C::SUPER && round < 3 ? to[q] < WP_BLOCK && (to[q] >> 4) == (lane >> 4) : to[q] < WP_BLOCK;
            to[q] = inside ? further[q] : to[q];
          }
          if (C::SUPER && round == 2) {
#pragma unroll
            for (int q = 0; q < SIDE; ++q) {
              const uint32_t p = (b0 + q) * WP_BLOCK + lane;
              const uint32_t abs_to = to[q] >= WP_STOP ? (to[q] == WP_END ? N : to[q]) : (b0 + q) * WP_BLOCK + to[q];
              if (p <= N) L.tab16[p] = (uint16_t)abs_to;
            }
          }
        }
#pragma unroll
        for (int q = 0; q < SIDE; ++q) {
          const uint32_t p = (b0 + q) * WP_BLOCK + lane;
          const uint32_t abs_to = to[q] >= WP_STOP ? (to[q] == WP_END ? N : to[q]) : (b0 + q) * WP_BLOCK + to[q];
          if (p <= N) L.tab[p] = (uint16_t)abs_to;
        }
      }
    }
    __syncthreads();
    WP_STAMP();
    // ---- 3. the chain, block by block; where it stops short of the window's end, the window ends there
    if constexpr (!C::SUPER) {
      if (tid == 0) {
        uint32_t p = 0;
        while (p < N) {
          L.entry_of[p / WP_BLOCK] = (uint16_t)p;
          const uint32_t q = L.tab[p];
          if (q == WP_STOP) {  // the run that does not fit starts in this block: walk up to it
            for (;;) {
              const uint32_t lo = funnel_shr(pgw[p >> 2], pgw[(p >> 2) + 1], (p & 3u) * 8u);
              const WholeRun r = whole_run(lo, w, vb);
              if (p + r.len > N) break;
              p += r.len;
            }
            L.n_eff = p;
            p = N;
            break;
          }
          p = q;
        }
        if (p != N) L.ok = 0u;  // WP_COMPLEX met (a chain position beyond N cannot occur)
      }
    } else {
      // 256 bytes at a hop: where a chain that enters at a byte leaves its four blocks (at most three more exits to
      // follow; every thread takes its bytes side by side), one thread hops through the window, then a thread per 256
      // bytes notes where the chain enters each of its four blocks.
      for (uint32_t p0 = tid; p0 <= N; p0 += 4u * (uint32_t)C::THREADS) {  // (four bytes side by side: their look-ups overlap)
        uint32_t t[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
          const uint32_t p = p0 + u * (uint32_t)C::THREADS;
          t[u] = L.tab[p <= N ? p : N];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
          for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t p = p0 + u * (uint32_t)C::THREADS;
            const bool inside = t[u] < WP_STOP && t[u] / WP_SUPER == p / WP_SUPER && t[u] <= N;
            const uint32_t further = L.tab[inside ? t[u] : 0u];
            t[u] = inside ? further : t[u];
          }
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
          const uint32_t p = p0 + u * (uint32_t)C::THREADS;
          if (p <= N) L.tab2[p] = (uint16_t)t[u];
        }
      }
      if (tid <= C::BYTES / WP_SUPER) L.sentry[tid] = 0xFFFFu;
      __syncthreads();
      WP_STAMP();
      if (tid < 64u) {  // the first wave, all lanes alike: positions and loop control stay in scalar registers
        uint32_t p = 0;
        while (p < N) {
          L.sentry[p / WP_SUPER] = (uint16_t)p;
          const uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)L.tab2[p]);
          if (q == WP_STOP) {  // the run that does not fit starts in these 256 bytes: walk up to it
            for (;;) {
              const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)funnel_shr(pgw[p >> 2], pgw[(p >> 2) + 1], (p & 3u) * 8u));
              const WholeRun r = whole_run(lo, w, vb);
              if (p + r.len > N) break;
              p += r.len;
            }
            if (tid == 0) L.n_eff = p;
            p = N;
            break;
          }
          p = q;
        }
        if (p != N && tid == 0) L.ok = 0u;  // WP_COMPLEX met
      }
      __syncthreads();
      WP_STAMP();
      if (tid <= C::BYTES / WP_SUPER) {
        uint32_t p = L.sentry[tid];
        const uint32_t stop = (tid + 1u) * WP_SUPER;
        while (p < stop && p < N) {  // (0xFFFF: the chain does not come through here)
          L.entry_of[p / WP_BLOCK] = (uint16_t)p;
          p = L.tab[p];  // (a mark ends the loop: it is above every position)
        }
      }
      __syncthreads();
      if (tid < C::BYTES / WP_BLOCK) {  // ... and a thread per block where it enters each of the block's quarters
        uint32_t p = L.entry_of[tid];
        const uint32_t stop = (tid + 1u) * WP_BLOCK;
        while (p < stop && p < N) {
          L.entry16[p >> 4] = (uint16_t)p;
          p = L.tab16[p];
        }
      }
    }
    __syncthreads();
    if (!L.ok) WP_FAIL(4);
    const uint32_t n_eff = L.n_eff;  // the window's bytes that are whole runs
    if (n_eff == 0u) WP_FAIL(5);   // a run longer than a window
    WP_STAMP();
    // ---- 4. the runs of every block, numbered, with their first output value
    // (a thread per block; in the long shape a thread per quarter of a block)
    const uint32_t block_end = (tid + 1u) * (C::SUPER ? 16u : WP_BLOCK);
    uint32_t my_runs = 0, my_vals = 0;
    uint32_t entry = 0xFFFFu;
    if constexpr (C::SUPER) {
      if (tid < C::BYTES / 16u) entry = L.entry16[tid];
    } else {
      if (tid < (C::BYTES / WP_BLOCK)) entry = L.entry_of[tid];
    }
    if (entry != 0xFFFFu) {
      uint32_t p = entry;
      while (p < block_end && p < n_eff) {
        const uint32_t lo = funnel_shr(pgw[p >> 2], pgw[(p >> 2) + 1], (p & 3u) * 8u);
        const WholeRun r = whole_run(lo, w, vb);
        ++my_runs;
        my_vals += r.nvals < C::VALUES ? r.nvals : C::VALUES;
        p += r.len;
      }
    }
    uint32_t total_runs, total_vals;
    uint32_t run_i = wp_exclusive<C::THREADS>(my_runs, L.wave_totals, total_runs);
    uint32_t val_i = wp_exclusive<C::THREADS>(my_vals, L.wave_totals, total_vals);  // (a run counts C::VALUES at most: sums stay small)
    // "Reading past RLE/BitPacking stream.": the other kernel reports it
    if (last && total_vals < n_window) WP_FAIL(6);
    uint32_t nv = total_vals < n_window ? total_vals : n_window;  // values this window delivers
    // more values in the window's runs than it delivers, and the page wants them: the window ends at its value nv
    const bool cut = nv < n_values && nv < total_vals;
    uint32_t* runs = reinterpret_cast<uint32_t*>(L.tab);  // nobody reads the exits any more (barriers above)
    if (entry != 0xFFFFu) {
      const uint32_t vmask = vb >= 4u ? ~0u : (1u << (8u * vb)) - 1u;
      uint32_t p = entry;
      while (p < block_end && p < n_eff) {
        const uint32_t lo = funnel_shr(pgw[p >> 2], pgw[(p >> 2) + 1], (p & 3u) * 8u);
        const WholeRun r = whole_run(lo, w, vb);
        if (val_i < nv) {
          // table entry: bit 31 set = bit-packed, then bits 0..30 = C::BIAS + window bit of the run's first value - first
          // value * w (value v of the run sits at entry + v * w - C::BIAS); else the repeated value, its vb bytes as they
          // are (the reference does not mask them to the width)
          const uint32_t q = p + r.nb;
          const uint32_t rep = funnel_shr(pgw[q >> 2], pgw[(q >> 2) + 1], (q & 3u) * 8u) & vmask;
          const bool crossing = cut && r.nvals > nv - val_i;  // the one run that reaches past the window's last value
          if (crossing && r.packed) {
            // the next window starts with it (one that starts this window is longer than a window takes: cut_nv 0)
            L.cut_at = p;
            L.cut_nv = val_i;
            L.carry_n = 0u;
          } else {
            runs[run_i] = r.packed ? (0x80000000u | (C::BIAS + q * 8u - val_i * w)) : rep;
            if constexpr (IDENT && C::SUPER) reinterpret_cast<uint32_t*>(L.tab2)[run_i] = val_i;  // (the 256-byte exits are done with)
            atomicOr(&L.flags[val_i >> 5], 1u << (val_i & 31u));
            if (r.nvals >= n_values - val_i) L.consumed = base + p + r.len;  // the run that holds the page's last value: read whole (:95-99)
            if (crossing) {  // an RLE run: the next window starts behind it, its remaining values in front
              L.cut_at = p + r.len;
              L.cut_nv = nv;
              L.carry_n = r.nvals - (nv - val_i);
              L.carry_val = rep;
            }
          }
        } else if (cut && val_i == nv) {  // no run crosses: the next window starts with this one
          L.cut_at = p;
          L.cut_nv = nv;
          L.carry_n = 0u;
        }
        ++run_i;
        val_i += r.nvals < C::VALUES ? r.nvals : C::VALUES;
        p += r.len;
      }
    }
    __syncthreads();
    uint32_t next_base = base + n_eff;
    if (cut) {
      if (L.cut_at == ~0u) WP_FAIL(7);  // (cannot happen: some run crosses)
      nv = L.cut_nv;
      next_base = base + L.cut_at;
      carry_n = L.carry_n;
      carry_val = L.carry_val;
      if (nv == 0u) WP_FAIL(8);  // a bit-packed run of more values than a window takes
    }
    __syncthreads();
    WP_STAMP();
    // run starts before each 64-bit word of the bitmap
    uint32_t n_runs;  // runs that hold values of the window (the table's entries)
    {
      const uint32_t words = (nv + 63u) >> 6;
      constexpr uint32_t PER = (C::VALUES / 64u + (uint32_t)C::THREADS - 1u) / (uint32_t)C::THREADS;  // words per thread
      uint32_t c[PER], sum = 0;
#pragma unroll
      for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t wd = tid * PER + k;
        c[k] = wd < words ? (uint32_t)(__builtin_popcount(L.flags[2 * wd]) + __builtin_popcount(L.flags[2 * wd + 1])) : 0u;
        sum += c[k];
      }
      uint32_t before = wp_exclusive<C::THREADS>(sum, L.wave_totals, n_runs);
#pragma unroll
      for (uint32_t k = 0; k < PER; ++k) {
        L.wbase[tid * PER + k] = (uint16_t)before;
        before += c[k];
      }
    }
    __syncthreads();
    WP_STAMP();
    // ---- 5. values.  A window of few runs: the dictionary entry of every RLE run is fetched once, by the threads side
    // by side, and kept behind the table (a lookup per VALUE costs the address path a cycle per lane: 20000 of them are
    // most of this routine's time).  An id beyond the dictionary among them leaves the window to the value-by-value form.
    bool fast = n_runs <= C::VAL_RUNS && !(IDENT && C::SUPER);
    if (fast) {
      const uint32_t* runs_c = reinterpret_cast<const uint32_t*>(L.tab);
      uint32_t* vals = reinterpret_cast<uint32_t*>(L.tab) + C::VAL_RUNS;
      for (uint32_t r = tid; r < n_runs; r += (uint32_t)C::THREADS) {
        const uint32_t e = runs_c[r];
        if (e >> 31) continue;
        if (IDENT) {
          vals[r] = e;
        } else if (e >= d.dict_len) {
          L.rle_oob = 1u;
        } else if (ELEM == 4) {
          vals[r] = ((const PQD_GLOBAL uint32_t*)d.dict)[e];
        } else {
          reinterpret_cast<u32x2*>(vals)[r] = ((const PQD_GLOBAL u32x2_a4*)d.dict)[e];
        }
      }
      __syncthreads();
      fast = L.rle_oob == 0u;
    }
    WP_STAMP();
    if constexpr (IDENT && C::SUPER) {
      if (tid == 0) reinterpret_cast<uint32_t*>(L.tab2)[n_runs] = nv;  // where the last run's values end
      __syncthreads();
      wp_values_split<C>(L, d, nv, v_base, w, n_runs, vt, n_vt);
    } else if (fast)
      wp_values<ELEM, true, IDENT, C>(L, d, nv, v_base, w);
    else
      wp_values<ELEM, false, IDENT, C>(L, d, nv, v_base, w);
    __syncthreads();  // the window's LDS is done with
    WP_STAMP();
    v_base += nv;
    if (v_base >= n_page_values) break;
    if (last && !cut) WP_FAIL(9);  // (cannot happen: the last window delivers what is still wanted or has returned above)
    base = next_base;
  }
  __syncthreads();
  if (tid == 0 && part == 0) {
    const uint32_t oob = L.oob_at;
    const uint32_t used = S.consumed_all != ~0u ? S.consumed_all : S.prefix + L.consumed;
    pqd_page_result r = {oob < n_page_values ? PQD_DICT_OOB : PQD_OK, oob < n_page_values ? oob : n_page_values, used, 0u};
    *result = r;
  }
#ifdef PQD_PROF
  __syncthreads();
  if (tid == 0)  // development build: the page's first values are replaced by the cycles of each step of the first window
    for (int k = 0; k + 1 < n_stamp; ++k) ((PQD_GLOBAL uint32_t*)d.out)[k] = (uint32_t)(stamp[k + 1] - stamp[k]);
#endif
  return true;
}

}  // namespace pqd
