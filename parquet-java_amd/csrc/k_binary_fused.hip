// Dictionary-encoded BINARY pages decoded in one pass: ids -> dictionary entry -> offsets[n + 1] + value bytes.
//
// The reference's read loop for such a column is DictionaryValuesReader.readBytes
// (parquet-column/.../values/dictionary/DictionaryValuesReader.java:76-83): id = decoder.readInt(), then
// dictionary.decodeToBinary(id) (PlainValuesDictionary.java:88-118: a view into the dictionary page), whose bytes
// the caller copies.  The two-step form of this library does the same through memory (ids -> 8-byte views with the
// dictionary gather, views -> bytes with k_binary_pages): 8 bytes per value written and read again, and two
// gathers per value (the view, then the bytes it points to).  Here the dictionary is a table of 16-byte SLOTS,
// made once per dictionary: an entry of up to 12 bytes lies in its slot together with its length, so a value
// costs ONE gather, and nothing goes through memory in between; longer entries keep {offset, length} in the slot
// and take a second load.
//
// One 256-thread workgroup per page.  The id stream passes through LDS a window of 8 KiB at a time; one lane walks
// the window's run headers (rle_scan_window: every rule of the reference's decoder), all lanes then take the
// window's values in tiles of 1024, four consecutive values per lane: ids from the window, slots from memory (L2),
// a workgroup prefix sum of the lengths, the bytes OR-ed into a zeroed LDS window at their place, the window
// written with aligned 16-byte stores (as k_binary_pages does).
// Bound: the gather (a lane-divergent load costs the address path about two cycles per lane whatever its size);
// algorithmic bytes per value: w / 8 in, 4 (offset) + the value's length out.
#include <hip/hip_ext.h>

#include "pqd_device.hpp"
#include "pqd_kernels.hpp"
#include "pqd_rle_scan.hpp"

#ifndef BF_EXP
#define BF_EXP 0
#endif

namespace pqd {

constexpr int BF_THREADS = 256;
constexpr int BF_VPL = 4;  // values per lane and tile
constexpr uint32_t BF_TILE = BF_THREADS * BF_VPL;
#ifndef BF_STREAM_KIB
#define BF_STREAM_KIB 8
#endif
#ifndef BF_DEFER  // 1: a tile's last window leaves at the top of the next tile (two windows in LDS); 0: at once
#define BF_DEFER 1
#endif
#ifndef BF_PREFETCH  // 1: the next stream window is requested while this one's values are worked on
#define BF_PREFETCH 1
#endif
#ifndef BF_WAVES
#define BF_WAVES 4
#endif
constexpr int BF_STREAM = BF_STREAM_KIB * 1024;  // stream bytes in LDS at a time
constexpr int BF_SLACK = 48;
constexpr int BF_PIECES = (BF_STREAM + BF_SLACK) / 16;
constexpr int BF_PRE = (BF_PIECES + BF_THREADS - 1) / BF_THREADS;  // pieces of the next window a lane holds in registers
#ifndef BF_SEGS_N
#define BF_SEGS_N 256
#endif
constexpr int BF_SEGS = BF_SEGS_N;
constexpr uint32_t BF_WIN = 12 * 1024;  // output bytes staged per window: a tile of inline values
constexpr uint32_t BF_NONE = 0xFFFFFFFFu;

typedef uint32_t bf_u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));

// ORs `count` (1..16) bytes held in `v` into the zeroed window at byte position `dst`, eight bytes at a time.
// padded: the bytes of `v` behind `count` are zero already (an entry out of its slot).
__device__ __forceinline__ void bf_or_bytes(unsigned long long* win, uint32_t dst, u32x4 v, uint32_t count, bool padded) {
  if (!padded) v = keep_leading_bytes(v, (long)count);
  const uint64_t lo = ((uint64_t)v.y << 32) | v.x, hi = ((uint64_t)v.w << 32) | v.z;
  const uint32_t sh = (dst & 7u) * 8u, q = dst >> 3;
  const uint32_t nq = ((dst & 7u) + count + 7u) >> 3;  // quadwords touched: 1..3
  const uint64_t p0 = lo << sh;
  const uint64_t p1 = sh ? (lo >> (64u - sh)) | (hi << sh) : hi;
  const uint64_t p2 = sh ? hi >> (64u - sh) : 0ull;
  atomicOr(&win[q], (unsigned long long)p0);
  if (nq > 1u) atomicOr(&win[q + 1], (unsigned long long)p1);
  if (nq > 2u) atomicOr(&win[q + 2], (unsigned long long)p2);
}

// a + b, or 2^32 - 1 when that does not fit (lengths are summed in 32 bits; every sum that matters lies below
// the capacity, which is below 2^31)
__device__ __forceinline__ uint32_t sat_add(uint32_t a, uint32_t b) {
  const uint32_t s = a + b;
  return s < a ? 0xFFFFFFFFu : s;
}
__device__ __forceinline__ uint32_t wave_scan_inclusive_sat(uint32_t v) {
  v = sat_add(v, dpp_or_zero<0x111, 0xF>(v));
  v = sat_add(v, dpp_or_zero<0x112, 0xF>(v));
  v = sat_add(v, dpp_or_zero<0x114, 0xF>(v));
  v = sat_add(v, dpp_or_zero<0x118, 0xF>(v));
  v = sat_add(v, dpp_or_zero<0x142, 0xA>(v));
  v = sat_add(v, dpp_or_zero<0x143, 0xC>(v));
  return v;
}

struct BfLds {
  u32x4 stream[BF_PIECES];
  unsigned long long win[1 + BF_DEFER][(BF_WIN + 32) / 8];  // two output windows: one is filled while the other leaves
  uint32_t seg_start[BF_SEGS + 1];
  uint32_t seg_src[BF_SEGS];
  uint32_t seg_kind[BF_SEGS];
  uint32_t wave_total[BF_THREADS / 64];
  uint32_t stop_bytes;
  uint32_t stop, stop_kind, stop_flag;
  RleState st;
};

constexpr uint32_t BF_STOP_OOB = 1, BF_STOP_VIEW = 2, BF_STOP_ROOM = 3;

#if BF_EXP == 9  // development: cycles of every phase of a page, summed by thread 0, left in the page's first offsets
#define BF_T(k)                                         \
  {                                                     \
    const uint64_t now_ = __builtin_readcyclecounter(); \
    t_acc[k] += (uint32_t)(now_ - t_last);              \
    t_last = now_;                                      \
  }
#else
#define BF_T(k)
#endif

// A filled window -> memory, aligned 16-byte stores; the window is zero again afterwards.  The window's LDS byte 0
// stands for the 16-byte aligned address gdst - s; its bytes [s, s + wlen) are this window's, the bytes
// [head_from, s) in front of them came over from the window before, the bytes [0, head_from) are not the page's at
// all (head_from != 0: only while the page's first piece has not left).  The piece that holds the window's end, if
// it is not full: written byte by byte when nothing follows (final); else handed back to the caller (the lane that
// holds it gets carry_mine), who puts it in front of the next window, so that a page writes whole pieces all the
// way but for its two ends.  Returns head_from of the next window.
__device__ __forceinline__ uint32_t bf_flush(unsigned long long* win, PQD_GLOBAL uint8_t* gdst, uint32_t s, uint32_t wlen, uint32_t tid,
                                             uint32_t head_from, bool final, u32x4& carry, bool& carry_mine) {
  const uint32_t wend = s + wlen;
  const uint32_t pieces = (wend + 15u) >> 4;
  carry_mine = false;
  for (uint32_t i = tid; i < pieces; i += BF_THREADS) {
    const u32x4 v = ((const u32x4*)win)[i];
    ((u32x4*)win)[i] = u32x4{0u, 0u, 0u, 0u};
    const uint32_t p0 = i * 16u;
    const bool whole_head = i != 0u || head_from == 0u, whole_tail = p0 + 16u <= wend;
    if (whole_head && whole_tail) {
#if BF_EXP != 3  // development: 3 = no byte stores
      PQD_STREAM_STORE((PQD_GLOBAL u32x4*)(gdst - s + p0), v);
#endif
    } else if (!whole_tail && !final) {
      carry = v;
      carry_mine = true;
    } else {  // the page's first and last piece: only the bytes that belong to it
      const uint32_t from = i != 0u ? p0 : head_from;
#pragma unroll 1
      for (uint32_t k = 0; k < 16u; ++k) {
        const uint32_t word = k < 8u ? (k < 4u ? v.x : v.y) : (k < 12u ? v.z : v.w);
        if (p0 + k >= from && p0 + k < wend) gdst[(long)(p0 + k) - (long)s] = (uint8_t)(word >> ((k & 3u) * 8u));
      }
    }
  }
  return wend < 16u && !final ? head_from : 0u;  // (the first piece stays in LDS only when it is also the unfinished last one)
}

__global__ __launch_bounds__(BF_THREADS) __attribute__((amdgpu_waves_per_eu(BF_WAVES))) void k_binary_fused(
    const pqd_page_desc* __restrict__ pages, const pqd_binary_page_desc* __restrict__ bins, const u32x4* __restrict__ slots_,
    uint32_t n_slots, pqd_page_result* __restrict__ page_results, pqd_page_result* __restrict__ bin_results, uint32_t n_pages) {
  __shared__ BfLds L;
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const pqd_page_desc d = pages[page];
  const pqd_binary_page_desc b = bins[page];
  const PQD_GLOBAL u32x4* slots = (const PQD_GLOBAL u32x4*)slots_;
  const uint32_t* win_s = reinterpret_cast<const uint32_t*>(L.stream);
  const uint32_t n_values = d.num_values;
  // ---- what DictionaryValuesReader.initFromPage looks at (DictionaryValuesReader.java:49-63)
  int32_t early = PQD_OK;
  bool finished = false;
  uint32_t w = 0;
  const uint32_t end = d.data_len;
  if (d.encoding != PQD_ENC_RLE_DICTIONARY || d.elem_size != 8) {
    early = PQD_BAD_ARG;
    finished = true;
  } else if (end == 0) {
    early = n_values ? PQD_DECODE : PQD_OK;  // an empty page fails on the first read only
    finished = true;
  } else {
    w = ((const PQD_GLOBAL uint8_t*)d.data)[0];
    if (w > 32) {  // RunLengthBitPackingHybridDecoder.java:55
      early = PQD_BAD_ARG;
      finished = true;
    }
  }
  PQD_GLOBAL uint32_t* offsets = (PQD_GLOBAL uint32_t*)b.offsets;
  PQD_GLOBAL uint8_t* bytes = (PQD_GLOBAL uint8_t*)b.bytes;
  if (finished || n_values == 0) {
    if (tid == 0) {
      pqd_page_result r = {early, 0u, early == PQD_OK && end ? 1u : 0u, 0u};
      page_results[page] = r;
      pqd_page_result q = {early == PQD_BAD_ARG ? PQD_BAD_ARG : PQD_OK, 0u, 0u, 0u};
      bin_results[page] = q;
      if (early != PQD_BAD_ARG && offsets) offsets[0] = 0u;
    }
    return;
  }
  const uint32_t dict_len = d.dict_len < n_slots ? d.dict_len : n_slots;
  const uint64_t cap = b.bytes_capacity > 0x7fffffffu ? 0x7fffffffull : b.bytes_capacity;  // int32 offsets
  const PQD_GLOBAL uint8_t* dict_bytes = (const PQD_GLOBAL uint8_t*)b.dict_bytes;
  const uint64_t dict_bytes_len = b.dict_bytes_len;
  const PQD_GLOBAL uint8_t* always_there = (const PQD_GLOBAL uint8_t*)(page_results + page);  // 16 readable bytes
  for (uint32_t i = tid; i < (1u + BF_DEFER) * ((BF_WIN + 32) / 16); i += BF_THREADS) ((u32x4*)L.win)[i] = u32x4{0u, 0u, 0u, 0u};
  if (tid == 0) {
    L.stop = BF_NONE;
    L.stop_flag = 0;
    L.st.pos = 1;
    L.st.produced = 0;
    L.st.window_values = 0;
    L.st.nseg = 0;
    L.st.pending_groups = 0;
    L.st.scan_status = PQD_OK;
    L.st.scan_err_at = ~0u;
    L.st.oob_at = ~0u;
    L.st.done = 0;
    L.st.window_base = 0;
  }
  const PQD_GLOBAL uint8_t* gend = (const PQD_GLOBAL uint8_t*)d.data + end;
  const uint32_t mask = mask32(w);
  uint32_t running = 0;    // bytes placed so far
  uint32_t delivered = 0;  // values materialised so far
  uint32_t stop_kind = 0;
  bool stopped = false;
  // the output window that was filled last and has not left yet (wlen 0: none)
  uint32_t cur_buf = 0, pend_buf = 0, pend_s = 0, pend_len = 0;
  PQD_GLOBAL uint8_t* pend_dst = bytes;
  uint32_t head_from = (uint32_t)((uintptr_t)bytes & 15u);  // see bf_flush: the page's bytes begin there in their first piece
  u32x4 carry = {0u, 0u, 0u, 0u};
  bool carry_mine = false;
#if BF_EXP == 9
  uint32_t t_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t t_last = __builtin_readcyclecounter();
#endif
  // the first window of the stream, requested here and put into LDS at the top of the loop
  u32x4 pre[BF_PRE];
  const PQD_GLOBAL uint8_t* pre_base;
#define BF_REQUEST_WINDOW(pos_)                                                                             \
  {                                                                                                         \
    const PQD_GLOBAL uint8_t* gptr_ = (const PQD_GLOBAL uint8_t*)d.data + (pos_);                           \
    pre_base = (const PQD_GLOBAL uint8_t*)((uintptr_t)gptr_ & ~(uintptr_t)15);                              \
    _Pragma("unroll") for (int k = 0; k < BF_PRE; ++k) {                                                    \
      const PQD_GLOBAL uint8_t* p_ = pre_base + (size_t)(tid + k * BF_THREADS) * 16;                        \
      pre[k] = u32x4{0u, 0u, 0u, 0u};                                                                       \
      if (tid + k * BF_THREADS < (uint32_t)BF_PIECES && p_ < gend) pre[k] = *(const PQD_GLOBAL u32x4*)p_;   \
    }                                                                                                       \
  }
#if BF_PREFETCH
  BF_REQUEST_WINDOW(1u)
#endif
  uint32_t pre_pos = 1;  // the stream position the requested window starts at
  __syncthreads();

  for (;;) {
    BF_T(9)
    // ---- the next window of the stream: out of the registers (bytes behind the stream's end are zero)
    const uint32_t cur = pre_pos;
#if !BF_PREFETCH
    BF_REQUEST_WINDOW(cur)
#endif
    const int32_t wbase = (int32_t)cur - (int32_t)(((const PQD_GLOBAL uint8_t*)d.data + cur) - pre_base);
#pragma unroll
    for (int k = 0; k < BF_PRE; ++k) {
      const uint32_t i = tid + k * BF_THREADS;
      if (i < (uint32_t)BF_PIECES) L.stream[i] = keep_leading_bytes(pre[k], (long)(gend - (pre_base + (size_t)i * 16)));
    }
    __syncthreads();
    BF_T(0)
    // The run headers.  What a writer makes of ids that do not repeat is one bit-packed run of 63 groups (504 values,
    // header byte 0x7F) after the other: the lanes of wave 0 look at the places where the headers of such runs would be,
    // all at once (behind what is left of a run that began in the window before), and the runs in front of the first
    // place that holds something else, or whose run is not wholly in the window, in the stream and wanted, are in the
    // table without a walk.  Then one lane walks on: ordinary runs (one header byte, wholly in the window and wanted)
    // in a short loop, and whatever that loop does not take with the routine that knows every rule
    // (rle_scan_window begins behind what is in the table and treats those runs exactly so).
    uint32_t nseg0 = 0, vals0 = 0;
    if (wave == 0 && w != 0u) {
      const uint32_t pend = L.st.pending_groups;
      const uint32_t remaining = n_values - L.st.produced;
      const int64_t lim = (int64_t)wbase + BF_STREAM < (int64_t)end ? (int64_t)wbase + BF_STREAM : (int64_t)end;
      const uint64_t pend_bytes = (uint64_t)pend * w;
      const bool head_ok = pend == 0u || ((int64_t)(cur + pend_bytes) <= lim && (uint64_t)pend * 8u <= remaining);
      if (head_ok) {
        const uint32_t run_bytes = 1u + 63u * w;
        const uint32_t base_vals = pend * 8u, seg0 = pend ? 1u : 0u;
        const uint64_t p = cur + pend_bytes + (uint64_t)lane * run_bytes;
        bool ok = (int64_t)(p + run_bytes) <= lim && (uint64_t)base_vals + (lane + 1u) * 504u <= remaining;
        ok = ok && lds_byte(win_s, (uint32_t)((int64_t)p - wbase)) == 0x7Fu;
        const uint64_t all = __builtin_amdgcn_ballot_w64(ok);
        const uint32_t n_reg = all == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~all);
        if (lane < n_reg) {
          L.seg_start[seg0 + lane] = base_vals + lane * 504u;
          L.seg_src[seg0 + lane] = (uint32_t)((int64_t)p + 1 - wbase);
          L.seg_kind[seg0 + lane] = SEG_PACKED;
        }
        nseg0 = seg0 + n_reg;
        vals0 = base_vals + n_reg * 504u;
        if (lane == 0) {
          if (pend) {
            L.seg_start[0] = 0u;
            L.seg_src[0] = (uint32_t)((int32_t)cur - wbase);
            L.seg_kind[0] = SEG_PACKED;
            L.st.pending_groups = 0u;
          }
          // ordinary runs, one after the other
          uint32_t pos = cur + (uint32_t)pend_bytes + n_reg * run_bytes, nseg = nseg0, vals = vals0;
          const uint32_t vb = (w + 7u) >> 3;
          const uint32_t vmask = vb >= 4u ? ~0u : (1u << (8u * vb)) - 1u;
          const int64_t wend = (int64_t)wbase + BF_STREAM;
          while (nseg < (uint32_t)BF_SEGS - 1u && (int64_t)pos + 8 <= lim && vals < remaining) {
            if (nseg > 0u && (int64_t)pos + 16 > wend && wend < (int64_t)end) break;  // (as the routine below: a new window first)
            const uint32_t o = (uint32_t)((int32_t)pos - wbase), i = o >> 2, sh = (o & 3u) * 8u;
            const uint32_t d0 = win_s[i], d1 = win_s[i + 1], d2 = win_s[i + 2];
            const uint32_t head = funnel_shr(d0, d1, sh), next4 = funnel_shr(d1, d2, sh);
            const uint32_t hb = head & 0xFFu;
            if (hb & 0x80u) break;
            const uint32_t cnt = hb >> 1;
            if (cnt == 0u) break;
            if (hb & 1u) {
              const uint32_t len = 1u + cnt * w, nv = cnt * 8u;
              if ((int64_t)pos + len > lim || nv > remaining - vals) break;
              L.seg_start[nseg] = vals;
              L.seg_src[nseg] = o + 1u;
              L.seg_kind[nseg] = SEG_PACKED;
              vals += nv;
              pos += len;
            } else {
              if (cnt > remaining - vals) break;
              L.seg_start[nseg] = vals;
              L.seg_src[nseg] = ((head >> 8) | (next4 << 24)) & vmask;  // not masked to the width (BytesUtils.java:124-142)
              L.seg_kind[nseg] = SEG_RLE;
              vals += cnt;
              pos += 1u + vb;
            }
            ++nseg;
          }
          nseg0 = nseg;
          vals0 = vals;
          L.st.pos = pos;
        }
      }
    }
    if (tid == 0) {
      L.st.window_base = wbase;
      rle_scan_window<BF_STREAM, BF_SEGS>(L.st, win_s, d.data, L.seg_start, L.seg_src, L.seg_kind, w, end, n_values, nseg0, vals0);
    }
    __syncthreads();
    BF_T(1)
    const uint32_t produced = L.st.produced;
    const uint32_t wv = L.st.window_values;
    const uint32_t nseg = L.st.nseg;
    const bool last_window = L.st.done != 0;
    if (!last_window) {  // the window after this one: on its way while this one's values are worked on
      pre_pos = L.st.pos;
#if BF_PREFETCH
      BF_REQUEST_WINDOW(pre_pos)
#endif
    }
    // the ids of a tile's values (lx: the lane's first value, counted from the window's first)
    auto tile_ids = [&](uint32_t lx, uint32_t (&id)[BF_VPL]) {
      id[0] = id[1] = id[2] = id[3] = 0u;
        if (lx < wv) {
          uint32_t a = 0, e = nseg;  // invariant: seg_start[a] <= lx < seg_start[e]
          while (e - a > 1) {
            const uint32_t m = (a + e) >> 1;
            if (L.seg_start[m] <= lx)
              a = m;
            else
              e = m;
          }
          const uint32_t s_begin = L.seg_start[a], s_end = L.seg_start[a + 1];
          if (lx + BF_VPL <= s_end) {
            const uint32_t src = L.seg_src[a];
            if (L.seg_kind[a] == SEG_PACKED && w <= 16u) {  // four ids in 64 bits behind any bit of a dword
              const uint32_t bit = src * 8 + (lx - s_begin) * w, i = bit >> 5, sh = bit & 31u;
              const uint32_t d0 = win_s[i], d1 = win_s[i + 1], d2 = win_s[i + 2];
              const uint64_t bits = ((uint64_t)funnel_shr(d1, d2, sh) << 32) | funnel_shr(d0, d1, sh);
              id[0] = (uint32_t)bits & mask;
              id[1] = (uint32_t)(bits >> w) & mask;
              id[2] = (uint32_t)(bits >> (2u * w)) & mask;
              id[3] = (uint32_t)(bits >> (3u * w)) & mask;
            } else if (L.seg_kind[a] == SEG_PACKED) {
              Window5 wnd = load_window(win_s, src * 8 + (lx - s_begin) * w);
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
            uint32_t t = a;
#pragma unroll
            for (int j = 0; j < BF_VPL; ++j) {
              const uint32_t l = lx + j;
              if (l >= wv) continue;
              while (L.seg_start[t + 1] <= l) ++t;
              const uint32_t src = L.seg_src[t];
              if (L.seg_kind[t] == SEG_PACKED) {
                const uint32_t bit = src * 8 + (l - L.seg_start[t]) * w;
                id[j] = funnel_shr(win_s[bit >> 5], win_s[(bit >> 5) + 1], bit & 31u) & mask;
              } else {
                id[j] = src;
              }
            }
          }
        }
    };
    uint32_t id_next[BF_VPL] = {0u, 0u, 0u, 0u};
    bool have_next = false;
    for (uint32_t t0 = 0; t0 < wv && !stopped; t0 += BF_TILE) {
      const uint32_t l0 = t0 + tid * BF_VPL;  // the lane's first value, counted from the window's first
      const uint32_t n_tile = wv - t0 < BF_TILE ? wv - t0 : BF_TILE;
      // ---- ids: worked out while the tile before this one waited for its slots, but for a window's first tile
      uint32_t id[BF_VPL];
      if (have_next) {
#pragma unroll
        for (int j = 0; j < BF_VPL; ++j) id[j] = id_next[j];
      } else {
        tile_ids(l0, id);
      }
      BF_T(2)
      BF_T(7)
      // ---- slots: one 16-byte gather per value, all four in flight together
      uint32_t len[BF_VPL], off[BF_VPL];
      uint32_t oob_mask = 0, bad_mask = 0, long_mask = 0, fast_mask = 0;
      u32x4 data[BF_VPL];
#pragma unroll
      for (int j = 0; j < BF_VPL; ++j) {
        const bool live = l0 + j < wv;
        const bool oob = live && id[j] >= dict_len;
        if (oob) oob_mask |= 1u << j;
        const PQD_GLOBAL u32x4* src = live && !oob ? slots + id[j] : (const PQD_GLOBAL u32x4*)nullptr;
#if BF_EXP == 1  // development: no gather
        if (src) {
          data[j] = u32x4{id[j], id[j] * 3u, id[j] * 5u, 4u + (id[j] & 7u)};
        } else {
#else
        if (src) {
          data[j] = *src;
        } else {
#endif
          const bf_u32x4_a1 raw = *(const PQD_GLOBAL bf_u32x4_a1*)always_there;
          data[j] = u32x4{raw.x, raw.y, raw.z, 0u};
        }
      }
      // ---- while the slots are on their way: the next tile's ids, and the window of the tile before this one leaves
      have_next = t0 + BF_TILE < wv;
      if (have_next) tile_ids(l0 + BF_TILE, id_next);
      if (pend_len) {  // (the window it filled last; the tile at hand fills the other one, whose first piece is free)
        head_from = bf_flush(L.win[pend_buf], pend_dst, pend_s, pend_len, tid, head_from, false, carry, carry_mine);
        if (carry_mine) ((u32x4*)L.win[cur_buf])[0] = carry;
        pend_s = (pend_s + pend_len) & 15u;  // what is left in LDS: the bytes in front of the next window
        pend_dst += pend_len;
        pend_len = 0;
        pend_buf = cur_buf;
      }
#pragma unroll
      for (int j = 0; j < BF_VPL; ++j) {
        const bool live = l0 + j < wv && !((oob_mask >> j) & 1u);
        const uint32_t tag = data[j].w;
        len[j] = 0;
        off[j] = 0;
        if (!live) continue;
        if (tag <= 12u) {  // the entry lies in the slot
          len[j] = tag;
          data[j].w = 0u;
          if (tag) fast_mask |= 1u << j;
        } else if (tag == 0x80000000u) {  // {offset, length} of an entry of more than 12 bytes
          off[j] = data[j].x;
          len[j] = data[j].y;
          long_mask |= 1u << j;
          if ((uint64_t)off[j] + len[j] > dict_bytes_len) {  // (pqd_dict_create never makes such a slot)
            bad_mask |= 1u << j;
            len[j] = 0;
          }
        } else {  // an entry whose length lies outside its dictionary page
          bad_mask |= 1u << j;
        }
      }
      if (__builtin_amdgcn_ballot_w64(long_mask != 0)) {  // entries of 13..16 bytes: one more 16-byte load
#pragma unroll
        for (int j = 0; j < BF_VPL; ++j) {
          const bool one_load = ((long_mask >> j) & 1u) && len[j] != 0 && len[j] <= 16u && (uint64_t)off[j] + 16u <= dict_bytes_len;
          if (one_load) {
            const bf_u32x4_a1 raw = *(const PQD_GLOBAL bf_u32x4_a1*)(dict_bytes + off[j]);
            data[j] = u32x4{raw.x, raw.y, raw.z, raw.w};
            fast_mask |= 1u << j;
          }
        }
      }
      BF_T(3)
      // ---- prefix sums of the lengths: lane, wave, workgroup (32 bits, saturating)
      uint32_t lane_total = 0;
#pragma unroll
      for (int j = 0; j < BF_VPL; ++j) lane_total = sat_add(lane_total, len[j]);
      const uint32_t inc = wave_scan_inclusive_sat(lane_total);
      if (lane == 63) L.wave_total[wave] = inc;
      if ((oob_mask | bad_mask) != 0) L.stop_flag = 1u;  // (never cleared: such a value ends the page in this tile)
      __syncthreads();
      // (inc - lane_total is exact unless inc saturated, and then the lane lies behind the value that ends the page)
      uint32_t lane_base = inc - lane_total, tile_total = 0;
#pragma unroll
      for (int k = 0; k < BF_THREADS / 64; ++k) {
        const uint32_t wt = L.wave_total[k];
        if ((uint32_t)k < wave) lane_base = sat_add(lane_base, wt);
        tile_total = sat_add(tile_total, wt);
      }
      uint32_t pre_sum[BF_VPL];
      {
        uint32_t acc = lane_base;
#pragma unroll
        for (int j = 0; j < BF_VPL; ++j) {
          pre_sum[j] = acc;
          acc = sat_add(acc, len[j]);
        }
      }
      uint32_t tile_bytes = tile_total;
      uint32_t tile_values = n_tile;
      uint32_t stop = BF_NONE;  // tile-relative, BF_NONE if the whole tile is delivered
      if (L.stop_flag != 0u || (uint64_t)running + tile_total > cap) {  // (uniform) the page ends in this tile
        // the first value that is not delivered: an id beyond the dictionary, a broken entry, bytes beyond the capacity
        uint64_t at = (uint64_t)running + lane_base;  // (a saturated base lies beyond the capacity as the true one does)
        uint32_t my_stop = BF_NONE;
#pragma unroll
        for (int j = 0; j < BF_VPL; ++j) {
          const bool fails = (((oob_mask | bad_mask) >> j) & 1u) || at + len[j] > cap;
          if (my_stop == BF_NONE && l0 + j < wv && fails) my_stop = tid * BF_VPL + j;
          at += len[j];
        }
        if (my_stop != BF_NONE) atomicMin(&L.stop, my_stop);
        __syncthreads();
        stop = L.stop;
        if (stop / BF_VPL == tid) {
          uint32_t at2 = lane_base;  // (exact: everything in front of the value that ends the page fits the capacity)
#pragma unroll
          for (int j = 0; j < BF_VPL; ++j) {
            if ((uint32_t)j == stop % BF_VPL) {
              L.stop_bytes = at2;
              L.stop_kind = ((oob_mask >> j) & 1u) ? BF_STOP_OOB : ((bad_mask >> j) & 1u) ? BF_STOP_VIEW : BF_STOP_ROOM;
            }
            at2 += len[j];
          }
        }
        __syncthreads();
        tile_bytes = L.stop_bytes;
        stop_kind = L.stop_kind;
        tile_values = stop;
        stopped = true;
      }
      BF_T(4)
      // ---- where the values start (the value that stops the page gets the end of the one before it)
      {
        const uint32_t g0 = produced + l0;
        if (l0 + BF_VPL <= wv && (stop == BF_NONE || tid * BF_VPL + BF_VPL - 1 <= stop)) {
          const u32x4_a4 o4 = {running + pre_sum[0], running + pre_sum[1], running + pre_sum[2], running + pre_sum[3]};
          PQD_STREAM_STORE((PQD_GLOBAL u32x4_a4*)(offsets + g0), o4);
        } else {
#pragma unroll
          for (int j = 0; j < BF_VPL; ++j)
            if (l0 + j < wv && tid * BF_VPL + j <= stop) offsets[g0 + j] = running + pre_sum[j];
        }
      }
      BF_T(5)
      // ---- the bytes, one LDS window after the other (one window unless the tile holds long entries); the
      // tile's last window stays in LDS and leaves at the top of the next tile
      for (uint32_t wb = 0; wb < tile_bytes; wb += BF_WIN) {
        const uint32_t wlen = tile_bytes - wb < BF_WIN ? tile_bytes - wb : BF_WIN;
        PQD_GLOBAL uint8_t* gdst = bytes + running + wb;
        const uint32_t s = (uint32_t)((uintptr_t)gdst & 15u);
        unsigned long long* win = L.win[cur_buf];
        uint32_t slow_mask = 0;
#pragma unroll
        for (int j = 0; j < BF_VPL; ++j) {
          if (tid * BF_VPL + j >= stop || len[j] == 0) continue;
          const uint32_t b0 = pre_sum[j], b1 = b0 + len[j];
          if (b1 <= wb || b0 >= wb + wlen) continue;
          if (((fast_mask >> j) & 1u) && b0 >= wb && b1 <= wb + wlen) {
#if BF_EXP != 2  // development: 2 = no ORs
            bf_or_bytes(win, s + (b0 - wb), data[j], len[j], !((long_mask >> j) & 1u));
#endif
          } else {
            slow_mask |= 1u << j;
          }
        }
        while (slow_mask) {  // byte by byte: entries of more than 16 bytes, the dictionary's last bytes, window edges
          const uint32_t j = (uint32_t)__builtin_ctz(slow_mask);
          slow_mask &= slow_mask - 1u;
          uint32_t b0 = 0, l = 0, src0 = 0, inl = 0;
          u32x4 dj = {0u, 0u, 0u, 0u};
#pragma unroll
          for (int k = 0; k < BF_VPL; ++k) {
            const bool me = (uint32_t)k == j;
            b0 = me ? pre_sum[k] : b0;
            l = me ? len[k] : l;
            src0 = me ? off[k] : src0;
            inl = me ? (uint32_t)!((long_mask >> k) & 1u) : inl;
            dj = me ? data[k] : dj;
          }
          const uint32_t lo = b0 > wb ? b0 : wb, hi = b0 + l < wb + wlen ? b0 + l : wb + wlen;
          uint32_t p = lo;
          if (!inl) {  // a long entry: 16 bytes at a time while 16 bytes can be read from the dictionary page
            for (; p < hi && (uint64_t)src0 + (p - b0) + 16u <= dict_bytes_len; p += 16u) {
              const bf_u32x4_a1 raw = *(const PQD_GLOBAL bf_u32x4_a1*)(dict_bytes + (uint64_t)src0 + (p - b0));
              bf_or_bytes(win, s + (p - wb), u32x4{raw.x, raw.y, raw.z, raw.w}, hi - p < 16u ? hi - p : 16u, false);
            }
            if (p > hi) p = hi;
          }
          for (; p < hi; ++p) {
            const uint32_t dst = s + (p - wb), k = p - b0;
            uint64_t byte;
            if (inl) {  // an inline entry that straddles a window edge: its bytes are in the registers
              const uint32_t word = k < 8u ? (k < 4u ? dj.x : dj.y) : dj.z;
              byte = (word >> ((k & 3u) * 8u)) & 0xFFu;
            } else {
              byte = dict_bytes[(uint64_t)src0 + k];
            }
            atomicOr(&win[dst >> 3], (unsigned long long)(byte << ((dst & 7u) * 8u)));
          }
        }
        __syncthreads();
        BF_T(6)
        if (!BF_DEFER || wb + BF_WIN < tile_bytes) {  // not the tile's last window: it leaves at once
          head_from = bf_flush(win, gdst, s, wlen, tid, head_from, false, carry, carry_mine);
          __syncthreads();
          if (carry_mine) ((u32x4*)win)[0] = carry;  // in front of the next window, which is filled in the same place
          pend_buf = cur_buf;
          pend_dst = gdst + wlen;
          pend_s = (s + wlen) & 15u;
          pend_len = 0;
          __syncthreads();
        } else {
          pend_buf = cur_buf;
          pend_dst = gdst;
          pend_s = s;
          pend_len = wlen;
          cur_buf ^= 1u;
        }
      }
      running += tile_bytes;
      delivered = produced + t0 + tile_values;
    }
    if (tid == 0) L.st.produced = produced + wv;
    if (last_window || stopped) break;
    __syncthreads();  // everybody is done with the stream window and the segment table
  }
  // what is still in LDS: the last window, or the bytes that came over from it into the next one's first piece
  if (pend_len || pend_s > head_from) bf_flush(L.win[pend_buf], pend_dst, pend_s, pend_len, tid, head_from, true, carry, carry_mine);

  if (tid == 0) {
    // the id stream's own result, as the dictionary decode reports it
    uint32_t limit = L.st.scan_status != PQD_OK ? L.st.scan_err_at : n_values;
    if (limit > n_values) limit = n_values;
    pqd_page_result r = {L.st.scan_status, limit, L.st.pos, 0u};
    pqd_page_result q = {PQD_OK, delivered, running, 0u};
    if (stopped) {
      r.values_written = delivered;
      if (stop_kind == BF_STOP_OOB) {
        r.status = PQD_DICT_OOB;
      } else {  // the values do not fit, or an entry is broken: the stream is not read further
        r.status = PQD_OK;
        q.status = stop_kind == BF_STOP_VIEW ? PQD_DECODE : PQD_BAD_ARG;
      }
    } else {
      offsets[delivered] = running;
    }
    page_results[page] = r;
    bin_results[page] = q;
#if BF_EXP == 9
    for (int k = 0; k < 10; ++k) offsets[k] = t_acc[k];
#endif
  }
}

hipError_t launch_binary_fused(const pqd_page_desc* pages, const pqd_binary_page_desc* bins, const void* slots, uint32_t n_slots,
                               pqd_page_result* page_results, pqd_page_result* bin_results, uint32_t n_pages, hipStream_t stream,
                               hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (n_pages == 0) return hipSuccess;
  if (ev_start && ev_stop)  // timed launch: the dispatch records the kernel's own start and end
    hipExtLaunchKernelGGL(k_binary_fused, dim3(n_pages), dim3(BF_THREADS), 0, stream, ev_start, ev_stop, 0, pages, bins, (const u32x4*)slots,
                          n_slots, page_results, bin_results, n_pages);
  else
    hipLaunchKernelGGL(k_binary_fused, dim3(n_pages), dim3(BF_THREADS), 0, stream, pages, bins, (const u32x4*)slots, n_slots, page_results,
                       bin_results, n_pages);
  return hipGetLastError();
}

}  // namespace pqd
