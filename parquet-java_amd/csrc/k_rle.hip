// RLE / bit-packing hybrid page decode, optionally fused with the dictionary gather.
//
// Replaces RunLengthBitPackingHybridDecoder.readInt/readNext
// (parquet-column/.../values/rle/RunLengthBitPackingHybridDecoder.java:52-109), its wrappers
// RunLengthBitPackingHybridValuesReader (:41-66, 4-byte length prefix) and DictionaryValuesReader
// (parquet-column/.../values/dictionary/DictionaryValuesReader.java:49-127, 1-byte width prefix,
// value = dictionary[id]).
//
// One 256-thread workgroup decodes one page:
//   1. the page bytes are staged into LDS, a window of WINDOW_BYTES at a time, with coalesced
//      16-byte loads (bytes past the end of the stream are staged as zero: a truncated final
//      bit-packed run decodes zeros, as in the Java);
//   2. lane 0 walks the run headers inside the window (varint header, then either the RLE value or
//      groups*w payload bytes) and writes a segment table {first output index, source} to LDS;
//   3. all lanes expand: each lane owns four consecutive output values per step, finds their
//      segment by binary search, extracts the ids from the LDS window with funnel shifts (or
//      broadcasts the RLE value), gathers dictionary[id] and stores 16 or 32 contiguous bytes.
// HBM-bound: w/8 bytes in + elem bytes out per value; the dictionary is read through L2.
#include "pqd_device.hpp"
#include "pqd_kernels.hpp"
#include "pqd_expand.hpp"
#include "pqd_rle_plan.hpp"
#include "pqd_rle_scan.hpp"
#include "pqd_whole_page.hpp"

namespace pqd {

constexpr int RLE_THREADS = 256;
constexpr int WINDOW_BYTES = 24 * 1024;
constexpr int WINDOW_SLACK = 48;  // >= 15 (alignment) + 20 (window over-read) rounded up
constexpr int WINDOW_PIECES = (WINDOW_BYTES + WINDOW_SLACK) / 16;
constexpr int MAX_SEGS = 512;

// The page decoder's LDS: a window of the stream and the table of its runs.  It shares its room with the whole-page
// routine's (a page is decoded by one or the other): 31 KB, and with the kernel's registers bounded to 96 five workgroups
// a CU (36.6 KB, 113-125 registers and four before; the run-heavy shapes follow the number of workgroups in flight:
// dictionary ids in short runs 541 -> 479 us, read as ints 580 -> 415 us; with three they took 814 us).
struct RlePageLds {
  u32x4 win[WINDOW_PIECES];
  uint32_t seg_start[MAX_SEGS + 1];
  uint32_t seg_src[MAX_SEGS];
  uint32_t seg_kind[MAX_SEGS];
  RleState st;
};

template <int ELEM, bool HAS_DICT>
__device__ __forceinline__ void emit4(const pqd_page_desc& d, uint32_t g0, uint32_t lo, uint32_t hi,
                                      const uint32_t id[4], uint32_t* oob_at) {
  // g0 = global output index of id[0]; only indices in [lo, hi) are written
  if (HAS_DICT) {
    const uint32_t dict_len = d.dict_len;
    uint32_t bad = ~0u;
#pragma unroll
    for (int j = 3; j >= 0; --j)
      if (g0 + j >= lo && g0 + j < hi && id[j] >= dict_len) bad = g0 + j;
    if (bad != ~0u) atomicMin(oob_at, bad);
    if (ELEM == 4) {
      const uint32_t* dict = reinterpret_cast<const uint32_t*>(d.dict);
      uint32_t v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = id[j] < dict_len ? dict[id[j]] : 0u;
      uint32_t* o = reinterpret_cast<uint32_t*>(d.out) + g0;
      if (g0 >= lo && g0 + 4 <= hi) {
        u32x4_a4 r = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<u32x4_a4*>(o) = r;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (g0 + j >= lo && g0 + j < hi) o[j] = v[j];
      }
    } else {
      const u32x2_a4* dict = reinterpret_cast<const u32x2_a4*>(d.dict);
      u32x2_a4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x2_a4 z = {0u, 0u};
        v[j] = id[j] < dict_len ? dict[id[j]] : z;
      }
      u32x2_a4* o = reinterpret_cast<u32x2_a4*>(d.out) + g0;
      if (g0 >= lo && g0 + 4 <= hi) {
        u32x4_a4 r0 = {v[0].x, v[0].y, v[1].x, v[1].y};
        u32x4_a4 r1 = {v[2].x, v[2].y, v[3].x, v[3].y};
        *reinterpret_cast<u32x4_a4*>(o) = r0;
        *reinterpret_cast<u32x4_a4*>(o + 2) = r1;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (g0 + j >= lo && g0 + j < hi) o[j] = v[j];
      }
    }
  } else {
    uint32_t* o = reinterpret_cast<uint32_t*>(d.out) + g0;
    if (g0 >= lo && g0 + 4 <= hi) {
      u32x4_a4 r = {id[0], id[1], id[2], id[3]};
      *reinterpret_cast<u32x4_a4*>(o) = r;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (g0 + j >= lo && g0 + j < hi) o[j] = id[j];
    }
  }
}

// One page, decoded by the whole workgroup.
template <int ELEM, bool HAS_DICT>
__device__ void rle_page(const pqd_page_desc* __restrict__ descs, pqd_page_result* __restrict__ results, uint32_t page,
                         RlePageLds& P) {  // P: LDS declared by the kernel
  u32x4* win4 = P.win;
  uint32_t* seg_start = P.seg_start;
  uint32_t* seg_src = P.seg_src;
  uint32_t* seg_kind = P.seg_kind;
  RleState& st = P.st;
  const uint32_t* win = reinterpret_cast<const uint32_t*>(win4);

  const pqd_page_desc d = descs[page];
  const uint32_t tid = threadIdx.x;

  // ---- validation and stream prefixes (uniform over the workgroup)
  int32_t early = PQD_OK;
  bool finished = false;
  uint32_t pos = 0, end = d.data_len, w = d.bit_width, consumed_override = ~0u;
  const uint32_t n_values = d.num_values;
  const bool is_dict = d.encoding == PQD_ENC_RLE_DICTIONARY;
  if ((d.encoding != PQD_ENC_RLE && !is_dict) || is_dict != HAS_DICT || (HAS_DICT && d.elem_size != ELEM) ||
      (!HAS_DICT && d.elem_size != 4)) {
    early = PQD_BAD_ARG;
    finished = true;
  } else if (is_dict || (d.flags & PQD_F_WIDTH_PREFIX)) {
    if (end == 0) {  // DictionaryValuesReader.java:56-63: an empty page fails on the first read only
      early = n_values ? PQD_DECODE : PQD_OK;
      finished = true;
    } else {
      w = d.data[0];
      pos = 1;
    }
  } else if (d.flags & PQD_F_LENGTH_PREFIX) {
    if (end < 4) {
      early = PQD_EOF;
      finished = true;
    } else {
      uint32_t len = (uint32_t)d.data[0] | ((uint32_t)d.data[1] << 8) | ((uint32_t)d.data[2] << 16) |
                     ((uint32_t)d.data[3] << 24);
      if ((int32_t)len < 0) {
        early = PQD_DECODE;
        finished = true;
      } else if (len > end - 4) {
        early = PQD_EOF;
        finished = true;
      } else {
        pos = 4;
        end = 4 + len;
        consumed_override = end;
      }
    }
  }
  if (!finished && w > 32) {  // RunLengthBitPackingHybridDecoder.java:55
    early = PQD_BAD_ARG;
    finished = true;
  }
  if (finished || n_values == 0) {
    if (tid == 0) {
      pqd_page_result r;
      r.status = early;
      r.values_written = 0;
      r.bytes_consumed = early == PQD_OK ? (consumed_override != ~0u ? consumed_override : pos) : 0;
      r.reserved = 0;
      results[page] = r;
    }
    return;
  }

  if (tid == 0) {
    st.pos = pos;
    st.produced = 0;
    st.window_values = 0;
    st.nseg = 0;
    st.pending_groups = 0;
    st.scan_status = PQD_OK;
    st.scan_err_at = ~0u;
    st.oob_at = ~0u;
    st.done = 0;
    st.window_base = 0;
  }
  __syncthreads();

  const uint8_t* gend = d.data + end;
  const uint32_t mask = mask32(w);
  for (;;) {
    // ---- 1. stage the window that starts at the 16-byte block holding st.pos
    const uint32_t cur = st.pos;
    const uint8_t* gptr = d.data + cur;
    const uint8_t* gbase = reinterpret_cast<const uint8_t*>((uintptr_t)gptr & ~(uintptr_t)15);
    const int32_t wbase = (int32_t)cur - (int32_t)(gptr - gbase);
    stage_window<RLE_THREADS>(win4, gbase, gend, WINDOW_PIECES);
    __syncthreads();
    // ---- 2. header walk
    if (tid == 0) {
      st.window_base = wbase;
      rle_scan_window<WINDOW_BYTES, MAX_SEGS>(st, win, d.data, seg_start, seg_src, seg_kind, w, end, n_values);
    }
    __syncthreads();
    // ---- 3. expansion
    const uint32_t produced = st.produced;
    const uint32_t wv = st.window_values;
    const uint32_t nseg = st.nseg;
    const uint32_t lo = produced, hi = produced + wv;
    if (wv > 0) {
      const uint32_t first_unit = lo >> 2, last_unit = (hi + 3) >> 2;
      uint32_t s = 0;  // segment cursor (units only move forward)
      for (uint32_t u = first_unit + tid; u < last_unit; u += RLE_THREADS) {
        const uint32_t g0 = u << 2;
        // local index of the first value of this unit that lies in the window
        const uint32_t l0 = (g0 > lo ? g0 : lo) - lo;
        // last segment with seg_start <= l0
        uint32_t a = s, b = nseg;  // invariant: seg_start[a] <= l0 < seg_start[b]
        while (b - a > 1) {
          uint32_t m = (a + b) >> 1;
          if (seg_start[m] <= l0)
            a = m;
          else
            b = m;
        }
        s = a;
        uint32_t id[4];
        const uint32_t s_begin = seg_start[s], s_end = seg_start[s + 1];
        if (g0 >= lo && g0 + 4 <= hi && (g0 - lo) + 4 <= s_end) {
          // whole unit inside one segment
          const uint32_t src = seg_src[s];
          if (seg_kind[s] == SEG_PACKED) {
            Window5 wnd = load_window(win, src * 8 + (g0 - lo - s_begin) * w);
            id[0] = wnd.d0 & mask;
            wnd.drop(w);
            id[1] = wnd.d0 & mask;
            wnd.drop(w);
            id[2] = wnd.d0 & mask;
            wnd.drop(w);
            id[3] = wnd.d0 & mask;
          } else {
            id[0] = id[1] = id[2] = id[3] = src;
          }
        } else {
          // unit straddles segments or the window edge: value by value
          uint32_t t = s;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            id[j] = 0;
            const uint32_t g = g0 + j;
            if (g < lo || g >= hi) continue;
            const uint32_t l = g - lo;
            while (seg_start[t + 1] <= l) ++t;
            const uint32_t src = seg_src[t];
            if (seg_kind[t] == SEG_PACKED) {
              const uint32_t bit = src * 8 + (l - seg_start[t]) * w;
              const uint32_t i = bit >> 5;
              id[j] = funnel_shr(win[i], win[i + 1], bit & 31u) & mask;
            } else {
              id[j] = src;
            }
          }
        }
        emit4<ELEM, HAS_DICT>(d, g0, lo, hi, id, &st.oob_at);
      }
    }
    __syncthreads();
    const bool done = st.done != 0;
    if (tid == 0) st.produced = produced + wv;
    if (done) break;
    __syncthreads();
  }

  if (tid == 0) {
    pqd_page_result r;
    uint32_t limit = st.scan_status != PQD_OK ? st.scan_err_at : n_values;
    if (limit > n_values) limit = n_values;
    if (HAS_DICT && st.oob_at < limit) {
      r.status = PQD_DICT_OOB;
      r.values_written = st.oob_at;
    } else {
      r.status = st.scan_status;
      r.values_written = limit;
    }
    r.bytes_consumed = consumed_override != ~0u ? consumed_override : st.pos;
    r.reserved = 0;
    results[page] = r;
  }
}

// Without a plan array: workgroup b decodes page b.  With one: only the pages the planned path could
// not take are decoded here; the plan kernel has marked them in their plan records (only_fallback[page].flags) and set
// the flag *fallback_count, and the grid strides over the records.
// Dictionary pages with a plan: the same launch then expands the short chunks the sliced gather leaves
// aside (pool != nullptr; pqd_expand.hpp), in the LDS the page decoder used.
template <int ELEM, bool HAS_DICT>
__global__ __launch_bounds__(RLE_THREADS, 5) void k_rle_pages(const pqd_page_desc* __restrict__ descs,
                                                           pqd_page_result* __restrict__ results,
                                                           uint32_t n_pages,
                                                           const RlePagePlan* __restrict__ only_fallback,
                                                           const uint32_t* __restrict__ fallback_count,
                                                           const RleSeg* __restrict__ pool, uint32_t pool_chunks,
                                                           const uint32_t* __restrict__ cursor, uint32_t* __restrict__ zero,
                                                           uint32_t zero_words) {
  constexpr int LDS_PIECES = (int)(((sizeof(WholePageLds) > sizeof(RlePageLds) ? sizeof(WholePageLds) : sizeof(RlePageLds)) + 15) / 16);
  __shared__ u32x4 win4[LDS_PIECES];  // the page decoder's window; the other parts of this kernel use it when no page is being decoded
  // the counter block of the NEXT call on this stream (nothing in this call reads or writes it)
  for (uint32_t i = blockIdx.x * RLE_THREADS + threadIdx.x; i < zero_words; i += gridDim.x * RLE_THREADS) zero[i] = 0u;
  static_assert(sizeof(u32x4) * WINDOW_PIECES >= sizeof(RleSeg) * SEG_CHUNK + 4 * (2 * EX_THREADS * EX_UNITS + 2), "the window holds the expansion's LDS");
  static_assert(RLE_THREADS == EX_THREADS, "one workgroup shape for both parts");
  if (!only_fallback) {
    if (blockIdx.x < n_pages) rle_page<ELEM, HAS_DICT>(descs, results, blockIdx.x, *reinterpret_cast<RlePageLds*>(win4));
    return;
  }
  // Pages the plan kernel left to this launch are marked in their plan records (PLAN_F_FALLBACK: for the page decoder;
  // PLAN_F_WHOLE: run-heavy dictionary pages, ids and lookups in one pass by dict_whole_page, and what that routine does
  // not cover goes to the page decoder too).  Whether there are any is a flag in the counter block (a count kept with
  // atomics made 20000 pages queue up for 200 us); workgroup b then looks at pages b, b + grid, ...
  bool any = __builtin_amdgcn_readfirstlane(*fallback_count) != 0u;
  if constexpr (HAS_DICT) any = any || (pool && __builtin_amdgcn_readfirstlane(cursor[PLAN_CTR_WHOLE]) != 0u);
  if (any) {
    static_assert(RLE_THREADS == WP_THREADS, "one workgroup shape for both parts");
    for (uint32_t page = blockIdx.x; page < n_pages; page += gridDim.x) {
      const pqd_page_desc d = descs[page];  // (requested together with the plan record: one trip to memory)
      const uint32_t f = __builtin_amdgcn_readfirstlane(only_fallback[page].flags) & ~PLAN_F_LONG;  // (a long page given up: the mark beside PLAN_F_FALLBACK)
      if (f != PLAN_F_FALLBACK && f != PLAN_F_WHOLE) continue;
      __syncthreads();  // the previous page's LDS is done with
      if constexpr (HAS_DICT) {
        if (f == PLAN_F_WHOLE && pool) {
          if (dict_whole_page<ELEM>(d, results + page, *reinterpret_cast<WholePageLds*>(win4))) continue;
          __syncthreads();
        }
      } else {
        if (f == PLAN_F_WHOLE) {  // run-heavy pages of dictionary ids read as ints
          if (dict_whole_page<4, true>(d, results + page, *reinterpret_cast<WholePageLds*>(win4))) continue;
          __syncthreads();
        }
      }
      rle_page<ELEM, HAS_DICT>(descs, results, page, *reinterpret_cast<RlePageLds*>(win4));
    }
    __syncthreads();
  }
  if constexpr (HAS_DICT) {
    if (pool) {
      RleSeg* cs_buf = reinterpret_cast<RleSeg*>(win4);
      uint32_t* slow_list = reinterpret_cast<uint32_t*>(cs_buf + SEG_CHUNK);
      dict_expand_direct_chunks<ELEM>(results, pool, pool_chunks, n_pages, cursor, cs_buf, slow_list, slow_list + 2 * EX_THREADS * EX_UNITS);
    }
  }
}

// Long pages of many short runs (a million values in 25000 runs), marked PLAN_F_LONG by the plan kernel: the whole-page
// routine in its long shape, a workgroup of sixteen waves per page.  What it cannot take goes to the left-over launch.
template <int ELEM, bool HAS_DICT>
__global__ __launch_bounds__(WpLong::THREADS) void k_whole_long_pages(const pqd_page_desc* __restrict__ descs,
                                                                      pqd_page_result* __restrict__ results,
                                                                      RlePagePlan* __restrict__ plans, uint32_t* __restrict__ cursor,
                                                                      uint32_t n_pages) {
  extern __shared__ u32x4 long_lds[];
  // plain values: two workgroups per page (each parses the page, each stores half of the values: a CU's stores are what
  // the value step waits for, and a call of 128 pages leaves half of the CUs free)
  constexpr uint32_t PARTS = HAS_DICT ? 1u : 2u;
  const uint32_t page = blockIdx.x / PARTS, part = blockIdx.x % PARTS;
  if (page >= n_pages) return;
  const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&plans[page].flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  // (the other part of the page may have finished already: both come to the same verdict, and write the same mark)
  if (f != PLAN_F_LONG && !(PARTS > 1u && (f == (PLAN_F_LONG | PLAN_F_DONE) || f == (PLAN_F_LONG | PLAN_F_FALLBACK)))) return;
  const pqd_page_desc d = descs[page];
  const bool done = dict_whole_page<ELEM, !HAS_DICT, WpLong>(d, results + page, *reinterpret_cast<WholePageLdsT<WpLong>*>(long_lds), part, PARTS);
  if (threadIdx.x == 0) {
    // PLAN_F_FALLBACK is what the left-over launch looks for; the long mark stays beside it for the page's other part
    __hip_atomic_store(&plans[page].flags, PLAN_F_LONG | (done ? PLAN_F_DONE : PLAN_F_FALLBACK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!done) cursor[PLAN_CTR_FALLBACK] = 1u;
  }
}

hipError_t launch_whole_long_pages(const pqd_page_desc* descs, pqd_page_result* results, void* plans, uint32_t* cursor, uint32_t n_pages,
                                   bool has_dict, int elem, hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  constexpr uint32_t LDS = (uint32_t)sizeof(WholePageLdsT<WpLong>);
  static_assert(LDS <= 160u * 1024u, "one workgroup's LDS");
  const void* fn = !has_dict ? reinterpret_cast<const void*>(k_whole_long_pages<4, false>)
                             : elem == 4 ? reinterpret_cast<const void*>(k_whole_long_pages<4, true>)
                                         : reinterpret_cast<const void*>(k_whole_long_pages<8, true>);
  static bool attr_set[3] = {false, false, false};  // (set again by a second thread at worst: the call is idempotent)
  const int which = !has_dict ? 0 : elem == 4 ? 1 : 2;
  if (!attr_set[which]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    if (e != hipSuccess) return e;
    attr_set[which] = true;
  }
  RlePagePlan* pl = reinterpret_cast<RlePagePlan*>(plans);
  const dim3 grid(has_dict ? n_pages : 2u * n_pages), block(WpLong::THREADS);
  if (!has_dict)
    hipLaunchKernelGGL((k_whole_long_pages<4, false>), grid, block, LDS, stream, descs, results, pl, cursor, n_pages);
  else if (elem == 4)
    hipLaunchKernelGGL((k_whole_long_pages<4, true>), grid, block, LDS, stream, descs, results, pl, cursor, n_pages);
  else
    hipLaunchKernelGGL((k_whole_long_pages<8, true>), grid, block, LDS, stream, descs, results, pl, cursor, n_pages);
  return hipGetLastError();
}

hipError_t launch_rle_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages, int elem,
                            bool has_dict, const void* only_fallback_plans, const uint32_t* fallback_count,
                            hipStream_t stream, const void* pool, uint32_t pool_chunks, const uint32_t* cursor, uint32_t* zero,
                            uint32_t zero_words) {
  if (n_pages == 0) return hipSuccess;
  const RlePagePlan* pl = reinterpret_cast<const RlePagePlan*>(only_fallback_plans);
  const RleSeg* sp = reinterpret_cast<const RleSeg*>(pool);
  // with a plan: a grid that strides over the left-over pages (few) and, for dictionary pages, over the direct chunks
  const uint32_t cap = 1280u;  // (five workgroups of this kernel fit a CU)
  const dim3 grid(pl && n_pages > cap ? cap : n_pages), block(RLE_THREADS);
  if (!has_dict)
    hipLaunchKernelGGL((k_rle_pages<4, false>), grid, block, 0, stream, descs, results, n_pages, pl, fallback_count, sp, pool_chunks, cursor, zero, zero_words);
  else if (elem == 4)
    hipLaunchKernelGGL((k_rle_pages<4, true>), grid, block, 0, stream, descs, results, n_pages, pl, fallback_count, sp, pool_chunks, cursor, zero, zero_words);
  else
    hipLaunchKernelGGL((k_rle_pages<8, true>), grid, block, 0, stream, descs, results, n_pages, pl, fallback_count, sp, pool_chunks, cursor, zero, zero_words);
  return hipGetLastError();
}

}  // namespace pqd
