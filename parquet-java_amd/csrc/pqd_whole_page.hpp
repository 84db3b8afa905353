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
// Covered: streams of at most 32 KiB (taken through LDS a window of 8 KiB at a time) that are at most half of what the
// page's values would take bit-packed, at most 32768 values, id widths 1..24, run headers of up to three bytes, runs
// shorter than a window.  Anything else, and every malformed stream (zero-length
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
constexpr uint32_t WP_BIAS = 1u << 20;  // above first value * w (< 2^15 * 24)
static_assert(WP_BLOCKS <= WP_THREADS, "a thread per block");

struct alignas(16) WholePageLds {
  uint8_t pg[WP_MAX_BYTES + 96];    // the stream, zeros behind it
  uint16_t tab[WP_MAX_BYTES + 64];  // where a chain entering at each byte leaves its block; later the run table (as words)
  uint32_t flags[WP_MAX_VALUES / 32 + 4];  // bit v: a run starts at value v
  uint16_t wbase[WP_MAX_VALUES / 64 + 4];  // run starts before each 64-bit word of the bitmap
  uint16_t entry_of[WP_BLOCKS];            // where the chain enters each block
  uint32_t wave_totals[WP_THREADS / 64];
  uint32_t ok, consumed, oob_at, rle_oob, n_eff;
};
static_assert((WP_MAX_BYTES / 2) * 4 <= sizeof(uint16_t) * (WP_MAX_BYTES + 64), "the run table fits where the exits were");

// Pages the plan kernel leaves to dict_whole_page (max_bytes: the launch's limit, at most WP_MAX_STREAM; 0: none).
__host__ __device__ inline bool whole_page_candidate(uint32_t stream_bytes, uint32_t n_values, uint32_t w, uint32_t dict_len,
                                                     uint32_t max_bytes) {
  return w >= 1u && w <= WP_MAX_WIDTH && n_values != 0u && n_values <= WP_MAX_VALUES && stream_bytes != 0u &&
         stream_bytes <= max_bytes && stream_bytes <= WP_MAX_STREAM && dict_len != 0u &&
         (uint64_t)stream_bytes * 16u <= (uint64_t)n_values * w;
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
__device__ __forceinline__ uint32_t wp_exclusive(uint32_t v, uint32_t* wave_totals, uint32_t& total) {
  const uint32_t inc = wave_scan_inclusive(v);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  __syncthreads();
  if (lane == 63) wave_totals[wave] = inc;
  __syncthreads();
  uint32_t before = 0;
  total = 0;
#pragma unroll
  for (uint32_t k = 0; k < WP_THREADS / 64; ++k) {
    const uint32_t t = wave_totals[k];
    before += k < wave ? t : 0u;
    total += t;
  }
  return before + inc - v;
}

// Where the looked-up values of the RLE runs sit when the page has at most WP_VAL_RUNS runs: behind that many table
// entries, in the space the block exits used.
constexpr uint32_t WP_VAL_RUNS = 1360;
static_assert(WP_VAL_RUNS * 12u <= sizeof(uint16_t) * (WP_MAX_BYTES + 64) && (WP_VAL_RUNS * 4u) % 8u == 0, "table and values fit where the exits were");

// The dictionary entries of values v0 .. v0 + 3 of the window (v0 a multiple of four below n_values, the window's number of
// values; v_base: the page's values before the window); x[k].y is
// meaningless for 4-byte dictionaries.  FAST: the entries of RLE runs were looked up once per run and are read from
// LDS; only values of bit-packed groups are fetched from the dictionary.
template <int ELEM, bool FAST, bool IDENT>
__device__ __forceinline__ void wp_unit(WholePageLds& L, const pqd_page_desc& d, uint32_t v0, uint32_t n_values, uint32_t v_base,
                                        uint32_t w, uint32_t mask, u32x2 x[4]) {
  const uint32_t* pgw = reinterpret_cast<const uint32_t*>(L.pg);
  const uint32_t* runs = reinterpret_cast<const uint32_t*>(L.tab);
  const uint32_t wd = v0 >> 6, sh = v0 & 63u;  // the four values share a word of the bitmap
  const uint64_t bits = (uint64_t)L.flags[2 * wd] | ((uint64_t)L.flags[2 * wd + 1] << 32);
  uint32_t ri = L.wbase[wd] + (uint32_t)__builtin_popcountll(bits & ((2ull << sh) - 1ull)) - 1u;
  const uint32_t near = (uint32_t)(bits >> sh);  // bits 1..3: a run starts at value v0 + 1..3
  uint32_t vw = v0 * w - WP_BIAS;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k) ri += (near >> k) & 1u;
    const uint32_t e = runs[ri];
    if (FAST && !(e >> 31)) {
      if (ELEM == 4)
        x[k].x = (runs + WP_VAL_RUNS)[ri];
      else
        x[k] = reinterpret_cast<const u32x2*>(runs + WP_VAL_RUNS)[ri];
    } else {
      const uint32_t bit = (e & 0x7FFFFFFFu) + vw;
      const uint32_t i0 = (bit >> 5) & (WP_MAX_BYTES / 4 - 1u);  // (an RLE entry gives a meaningless but harmless index)
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
template <int ELEM, bool FAST, bool IDENT>
__device__ __forceinline__ void wp_values(WholePageLds& L, const pqd_page_desc& d, uint32_t n_values, uint32_t v_base, uint32_t w) {
  const uint32_t mask = (1u << w) - 1u;
  constexpr uint32_t STEP = WP_THREADS * 4u;
  for (uint32_t v0 = threadIdx.x * 4u; v0 < n_values; v0 += 2u * STEP) {
    u32x2 x[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t vv = v0 + (uint32_t)u * STEP;
      if (vv < n_values) wp_unit<ELEM, FAST, IDENT>(L, d, vv, n_values, v_base, w, mask, x[u]);
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

// One dictionary page (d.data = [width][stream]) by the calling workgroup of WP_THREADS threads; L: its LDS, free for
// this call.  The stream is taken a window of WP_MAX_BYTES at a time; a window ends in front of the first run that does
// not fit into it.  false (for all threads): not decoded (what was written is written again by the caller's decoder).
// IDENT: a page of dictionary ids read as ints (PQD_ENC_RLE with PQD_F_WIDTH_PREFIX, DictionaryValuesReader.readValueDictionaryId):
// the same stream, the id itself is the value and no dictionary is looked at.
template <int ELEM, bool IDENT = false>
__device__ bool dict_whole_page(const pqd_page_desc& d, pqd_page_result* result, WholePageLds& L) {
  const uint32_t tid = threadIdx.x;
  const uint32_t n_page_values = d.num_values;
  if (d.data_len < 2u || (!IDENT && d.dict == nullptr)) return false;
  const uint32_t n_stream = d.data_len - 1u;
  if (n_stream > WP_MAX_STREAM || n_page_values == 0u || n_page_values > WP_MAX_VALUES) return false;
  // the width byte is requested here and looked at behind the first staging step: one trip to memory for both
  const uint32_t w = ((const PQD_GLOBAL uint8_t*)d.data)[0];
  if (tid == 0) {
    L.consumed = 0u;
    L.oob_at = ~0u;
  }
#ifdef PQD_PROF
  uint64_t stamp[10];
  int n_stamp = 0;
#define WP_STAMP() \
  if (base == 0u) stamp[n_stamp++] = __builtin_readcyclecounter()
#else
#define WP_STAMP()
#endif
  uint32_t base = 0;    // stream bytes before the window
  uint32_t v_base = 0;  // values before the window
  for (;;) {
    const bool last = n_stream - base <= WP_MAX_BYTES;
    const uint32_t N = last ? n_stream - base : WP_MAX_BYTES;
    const uint32_t n_values = n_page_values - v_base;  // values still wanted
    const PQD_GLOBAL uint8_t* src = (const PQD_GLOBAL uint8_t*)d.data + 1 + base;
    WP_STAMP();
    // ---- 1. the window -> LDS (zeros behind it).  Whole aligned dwords are read (none lies wholly outside the page) and
    // shifted into place: a piece of 16 stream bytes is five independent loads, whatever the page's address.
    {
      const uintptr_t a = (uintptr_t)src;
      const PQD_GLOBAL uint32_t* wsrc = (const PQD_GLOBAL uint32_t*)(a & ~(uintptr_t)3);
      const uint32_t off = (uint32_t)(a & 3u), sh = off * 8u;
      const uint32_t full = (off + N) >> 2;  // dwords wholly inside [aligned start, end of the window)
      const uint32_t tail = (off + N) & 3u;  // window bytes in dword `full`
      for (uint32_t i = tid * 4u; i * 4u < N + 16u; i += WP_THREADS * 4u) {
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
    for (uint32_t i = tid; i < (n_values >> 5) + 4u; i += WP_THREADS) L.flags[i] = 0u;
    if (tid < WP_BLOCKS) L.entry_of[tid] = 0xFFFFu;
    if (tid == 0) {
      L.ok = 1u;
      L.rle_oob = 0u;
      L.n_eff = N;
    }
    __syncthreads();
    if (!whole_page_candidate(n_stream, n_page_values, w, IDENT ? 1u : d.dict_len, WP_MAX_STREAM)) return false;
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
      for (uint32_t b0 = wave * SIDE; b0 < n_blocks; b0 += SIDE * (WP_THREADS / 64)) {
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
#pragma unroll
        for (int round = 0; round < 5; ++round) {
          uint32_t further[SIDE];
#pragma unroll
          for (int q = 0; q < SIDE; ++q) further[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((to[q] & 63u) << 2), (int)to[q]);
#pragma unroll
          for (int q = 0; q < SIDE; ++q) to[q] = to[q] < WP_BLOCK ? further[q] : to[q];
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
    __syncthreads();
    if (!L.ok) return false;
    const uint32_t n_eff = L.n_eff;  // the window's bytes that are whole runs
    if (n_eff == 0u) return false;   // a run longer than a window
    WP_STAMP();
    // ---- 4. the runs of every block, numbered, with their first output value
    const uint32_t block_end = (tid + 1u) * WP_BLOCK;
    uint32_t my_runs = 0, my_vals = 0;
    const uint32_t entry = tid < WP_BLOCKS ? L.entry_of[tid] : 0xFFFFu;
    if (entry != 0xFFFFu) {
      uint32_t p = entry;
      while (p < block_end && p < n_eff) {
        const uint32_t lo = funnel_shr(pgw[p >> 2], pgw[(p >> 2) + 1], (p & 3u) * 8u);
        const WholeRun r = whole_run(lo, w, vb);
        ++my_runs;
        my_vals += r.nvals < WP_MAX_VALUES ? r.nvals : WP_MAX_VALUES;
        p += r.len;
      }
    }
    uint32_t total_runs, total_vals;
    uint32_t run_i = wp_exclusive(my_runs, L.wave_totals, total_runs);
    uint32_t val_i = wp_exclusive(my_vals, L.wave_totals, total_vals);
    if (last && total_vals < n_values) return false;  // "Reading past RLE/BitPacking stream.": the other kernel reports it
    const uint32_t nv = total_vals < n_values ? total_vals : n_values;  // values this window delivers
    uint32_t* runs = reinterpret_cast<uint32_t*>(L.tab);  // nobody reads the exits any more (barriers above)
    if (entry != 0xFFFFu) {
      const uint32_t vmask = vb >= 4u ? ~0u : (1u << (8u * vb)) - 1u;
      uint32_t p = entry;
      while (p < block_end && p < n_eff) {
        const uint32_t lo = funnel_shr(pgw[p >> 2], pgw[(p >> 2) + 1], (p & 3u) * 8u);
        const WholeRun r = whole_run(lo, w, vb);
        if (val_i < nv) {
          // table entry: bit 31 set = bit-packed, then bits 0..30 = WP_BIAS + window bit of the run's first value - first
          // value * w (value v of the run sits at entry + v * w - WP_BIAS); else the repeated value, its vb bytes as they
          // are (the reference does not mask them to the width)
          const uint32_t q = p + r.nb;
          const uint32_t rep = funnel_shr(pgw[q >> 2], pgw[(q >> 2) + 1], (q & 3u) * 8u) & vmask;
          runs[run_i] = r.packed ? (0x80000000u | (WP_BIAS + q * 8u - val_i * w)) : rep;
          atomicOr(&L.flags[val_i >> 5], 1u << (val_i & 31u));
          const uint32_t rv = r.nvals < WP_MAX_VALUES ? r.nvals : WP_MAX_VALUES;
          if (val_i + rv >= n_values) L.consumed = base + p + r.len;  // the run that holds the page's last value: read whole (:95-99)
        }
        ++run_i;
        val_i += r.nvals < WP_MAX_VALUES ? r.nvals : WP_MAX_VALUES;
        p += r.len;
      }
    }
    __syncthreads();
    WP_STAMP();
    // run starts before each 64-bit word of the bitmap
    uint32_t n_runs;  // runs that hold values of the window (the table's entries)
    {
      const uint32_t words = (nv + 63u) >> 6;  // <= 512: two per thread
      uint32_t c[2], sum = 0;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const uint32_t wd = tid * 2u + k;
        c[k] = wd < words ? (uint32_t)(__builtin_popcount(L.flags[2 * wd]) + __builtin_popcount(L.flags[2 * wd + 1])) : 0u;
        sum += c[k];
      }
      uint32_t before = wp_exclusive(sum, L.wave_totals, n_runs);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        L.wbase[tid * 2u + k] = (uint16_t)before;
        before += c[k];
      }
    }
    __syncthreads();
    WP_STAMP();
    // ---- 5. values.  A window of few runs: the dictionary entry of every RLE run is fetched once, by the threads side
    // by side, and kept behind the table (a lookup per VALUE costs the address path a cycle per lane: 20000 of them are
    // most of this routine's time).  An id beyond the dictionary among them leaves the window to the value-by-value form.
    bool fast = n_runs <= WP_VAL_RUNS;
    if (fast) {
      const uint32_t* runs_c = reinterpret_cast<const uint32_t*>(L.tab);
      uint32_t* vals = reinterpret_cast<uint32_t*>(L.tab) + WP_VAL_RUNS;
      for (uint32_t r = tid; r < n_runs; r += WP_THREADS) {
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
    if (fast)
      wp_values<ELEM, true, IDENT>(L, d, nv, v_base, w);
    else
      wp_values<ELEM, false, IDENT>(L, d, nv, v_base, w);
    __syncthreads();  // the window's LDS is done with
    WP_STAMP();
    v_base += nv;
    if (v_base >= n_page_values) break;
    if (last) return false;  // (cannot happen: the last window delivers what is still wanted or has returned above)
    base += n_eff;
  }
  if (tid == 0) {
    const uint32_t oob = L.oob_at;
    pqd_page_result r = {oob < n_page_values ? PQD_DICT_OOB : PQD_OK, oob < n_page_values ? oob : n_page_values, 1u + L.consumed, 0u};
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
