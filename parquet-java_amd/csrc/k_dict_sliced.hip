// Dictionary-encoded pages, expansion + gather, for dictionaries larger than the L1 cache:
// the dictionary is streamed through LDS in slices and every lookup is an LDS read.
//
// Replaces the read loop of DictionaryValuesReader (parquet-column/.../values/dictionary/
// DictionaryValuesReader.java:75-118: value = dictionary.decodeToX(decoder.readInt())) for whole
// pages.  A random gather from a 256-512 KiB dictionary through the vector cache costs a cache
// line per value and runs at about 0.3 values/clk/CU; an LDS gather runs an order of magnitude
// faster, so instead of sending each id to the dictionary the dictionary is brought to the ids:
//
//   - one 1024-thread workgroup per CU walks over pages; the run plan of each page comes from
//     k_rle_plan, so nothing here is serial;
//   - a tile of up to 1024*R ids is unpacked into REGISTERS from packed dwords read straight from
//     global memory (coalesced; each lane owns units of four consecutive values);
//   - the dictionary streams through LDS in 64 KiB slices, double-buffered, by LDS-DMA
//     (global_load_lds_dwordx4) issued by all 16 waves;
//   - each pass resolves the ids that fall into the resident slice; a resolved id is replaced in
//     place by its value;
//   - while one page is being gathered, the next page's plan, segments and packed dwords are
//     already being fetched (behind passes 0 and 1), so no memory round trip is exposed between
//     pages;
//   - the values are stored once, 16 bytes per lane, fully coalesced.
// HBM traffic per value: w/8 bytes of ids + ELEM bytes out; the slices are L2 traffic
// (dictionary bytes per tile, i.e. a few bytes per value).  Small dictionaries (one slice) stay
// resident across pages.  Pages this kernel is not made for (more than SEG_CHUNK segments, a
// dictionary too large to sweep) are flagged by the plan kernel and left to the generic kernel.
#include "pqd_device.hpp"
#ifdef PQD_PROF
#include <cstdio>
#include <cstdlib>
#include <vector>
#endif
#include "pqd_kernels.hpp"
#include "pqd_rle_plan.hpp"

namespace pqd {

constexpr int DS_THREADS = 1024;
constexpr int DS_WAVES = DS_THREADS / 64;
constexpr int DS_R = DS_TILE_VALUES / DS_THREADS;  // values per lane per tile
constexpr int DS_SEG_BYTES = SEG_CHUNK * 8;         // one page's segments
constexpr int DS_PLAN_BYTES = 64;

constexpr int DS_NWP = 3;  // packed dwords kept per unit (enough for widths <= 16; wider: value by value)

constexpr int DS_OFF_SEGS = 2 * DS_SLICE_BYTES;               // three segment buffers
constexpr int DS_OFF_PLANS = DS_OFF_SEGS + 3 * DS_SEG_BYTES;  // three plan records
constexpr int DS_OFF_STATE = DS_OFF_PLANS + 4 * DS_PLAN_BYTES;
constexpr int DS_LDS_BYTES = DS_OFF_STATE + 64;
static_assert(DS_LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(DS_OFF_STATE % 16 == 0, "state block alignment");
static_assert(sizeof(RlePagePlan) == DS_PLAN_BYTES, "plan record size");
static_assert(DS_R % 4 == 0 && DS_R <= 32, "a lane holds whole units, one flag bit per value");

#define PQD_GLOBAL __attribute__((address_space(1)))
#define PQD_LDS __attribute__((address_space(3)))

// Build with -DPQD_PROF for per-phase cycle counts (development only; see the Makefile's prof target).
#ifdef PQD_PROF
#define PROF_INIT uint64_t pt_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; uint64_t pl_ = __builtin_readcyclecounter();
#define PROF(i) { const uint64_t n_ = __builtin_readcyclecounter(); pt_[i] += n_ - pl_; pl_ = n_; }
#define PROF_PARAM , uint64_t* prof
#define PROF_ARG , g_prof
#define PROF_FN_PARAM , uint64_t (&pt_)[12], uint64_t& pl_
#define PROF_FN_ARG , pt_, pl_
#else
#define PROF_INIT
#define PROF(i)
#define PROF_PARAM
#define PROF_ARG
#define PROF_FN_PARAM
#define PROF_FN_ARG
#endif

struct DsShared {
  uint32_t oob_at[3];  // first out-of-range id of the page in plan buffer 0 / 1 / 2
  uint32_t pad;
  uint32_t dummy[4];  // what lanes without a lookup read (never written after the prologue)
};

__device__ __forceinline__ uint32_t uni(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }

// LDS-DMA of 64 x 16 bytes: lane l's 16 bytes at gbase + goff land at LDS byte address
// lds_base + 16*l (gbase and lds_base wave-uniform).  Issued through inline assembly on purpose:
// the compiler orders every LDS-DMA it knows of against all later LDS accesses and LDS-DMAs with a
// full s_waitcnt vmcnt(0), which serialises the DMA instructions of a wave and stalls the lookups
// behind the slice being prefetched into the OTHER buffer.  Completion is awaited explicitly in
// wg_barrier().
__device__ __forceinline__ void dma16(const void* gbase, uint32_t goff, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base), "v"(goff), "s"(gbase)
               : "memory");
}

// Workgroup barrier that also waits for all of this wave's outstanding memory operations, the
// LDS-DMA included (the counters are in issue order).
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// The same, but the wave's N most recent memory operations may stay in flight.
template <int N>
__device__ __forceinline__ void wg_barrier_keep() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
// Barrier for LDS traffic only.
__device__ __forceinline__ void wg_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Fills one slice buffer; `bytes` is the valid length of the slice (a multiple of the element
// size; the dictionary is 16-byte aligned, the tail piece may read up to 15 bytes past it inside
// the same 16-byte block).
__device__ __forceinline__ void load_slice(const uint8_t* slice_src, uint32_t bytes, uint32_t lds_buf, uint32_t wave,
                                           uint32_t lane) {
  const uint32_t pieces = (bytes + 15) >> 4;
#pragma unroll
  for (int i = 0; i < (int)(DS_SLICE_BYTES / 16 / DS_THREADS); ++i) {
    const uint32_t piece0 = ((uint32_t)i * DS_WAVES + wave) * 64;  // wave-uniform
    if (piece0 + lane < pieces) dma16(slice_src, (piece0 + lane) * 16, uni(lds_buf + piece0 * 16));
  }
}

// A page's segments (1 KiB per wave, starting with wave 0; the segments of page p are pool chunk
// p) and plan record (the next wave) into the LDS buffers `seg_buf` / `plan_buf` (LDS byte
// addresses).
__device__ __forceinline__ void fetch_plan_async(const RlePagePlan* plans, const RleSeg* pool, uint32_t page,
                                                 uint32_t seg_buf, uint32_t plan_buf, uint32_t wave, uint32_t lane) {
  if (wave < (uint32_t)(DS_SEG_BYTES / 1024)) {
    dma16(pool + (size_t)page * SEG_CHUNK, wave * 1024 + lane * 16, uni(seg_buf + wave * 1024));
  } else if (wave == (uint32_t)(DS_SEG_BYTES / 1024)) {
    if (lane < (uint32_t)(DS_PLAN_BYTES / 16)) dma16(plans + page, lane * 16, uni(plan_buf));
  }
}

// Page constants, read from the plan record in LDS and held in scalar registers.
struct PageHead {
  const PQD_GLOBAL uint32_t* words;  // `data` rounded down to a dword
  uint32_t in_off;                   // data & 3
  uint32_t last_word;                // index of the last dword that holds a byte of the page
  uint32_t full_words;               // dwords lying wholly before the end of the stream
  uint32_t tail_mask;                // valid bytes of dword `full_words` (0: none)
  uint32_t n_out, w, nseg, flags;
};

__device__ __forceinline__ PageHead read_head(const PQD_LDS RlePagePlan* pl) {
  PageHead h;
  const uint32_t dlo = uni((uint32_t)pl->data), dhi = uni((uint32_t)(pl->data >> 32));
  const uint64_t data = ((uint64_t)dhi << 32) | dlo;
  h.in_off = dlo & 3u;
  h.words = (const PQD_GLOBAL uint32_t*)(data - h.in_off);
  const uint32_t data_len = uni(pl->data_len), stream_end = uni(pl->stream_end);
  h.last_word = data_len ? (h.in_off + data_len - 1) >> 2 : 0u;
  const uint32_t end_bytes = h.in_off + stream_end;
  h.full_words = end_bytes >> 2;
  h.tail_mask = (end_bytes & 3u) ? ((1u << (8 * (end_bytes & 3u))) - 1u) : 0u;
  h.n_out = uni(pl->n_out);
  h.w = uni(pl->w);
  h.nseg = uni(pl->nseg);
  h.flags = uni(pl->flags);
  return h;
}

// The packed dwords a unit of four values needs, fetched but not yet consumed.
template <int NW>
struct UnitRaw {
  uint32_t d[NW];
  uint32_t meta;  // bits 0..4: bit offset inside d[0]; bits 8..9: kind; bits 16..: segment index
};
constexpr uint32_t UK_NONE = 0, UK_PACKED = 1, UK_RLE = 2, UK_SLOW = 3;

__device__ __forceinline__ uint32_t seg_start(const PQD_LDS RleSeg* cs, uint32_t i) {
  return cs[i].start_kind & SEG_START_MASK;
}

// For every unit of the page's values [0, t1): find its segment and ISSUE the global loads of its packed
// dwords.  Nothing waits on the loads here, so all units' loads are in flight together (and, when
// this is called for the next page during a sweep, they complete behind the sweep).  NW = dwords
// kept per unit: units of wider ids (ceil((31 + 4w) / 32) > NW) are marked for the value-by-value
// path, as are units that straddle segments or lie in a segment cut short by the end of the stream.
template <int UNITS, int NW>
__device__ __forceinline__ void locate_fetch(const PQD_LDS RleSeg* cs, uint32_t count, uint32_t t1, const PageHead& h,
                                             uint32_t tid, UnitRaw<NW> (&raw)[UNITS] PROF_FN_PARAM) {
  constexpr uint32_t t0 = 0;
  const uint32_t nw = (62u + 4u * h.w) >> 5;
  // segment of every unit: a branch-free binary search, the units' steps interleaved so that
  // each step is one LDS round trip for all of them
  uint32_t seg[UNITS];
#pragma unroll
  for (int k = 0; k < UNITS; ++k) seg[k] = 0;
#pragma unroll
  for (uint32_t half = SEG_CHUNK / 2; half; half >>= 1) {
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
      const uint32_t g0 = t0 + 4 * ((uint32_t)k * DS_THREADS + tid);
      const uint32_t m = seg[k] + half;
      const uint32_t st = seg_start(cs, m < count ? m : count - 1);
      seg[k] = (m < count && st <= g0) ? m : seg[k];
    }
  }
  PROF(10)
  // Classification and addresses are plain arithmetic and the loads are issued by every lane for
  // every unit (units without packed dwords read the first dwords of the page): a load whose
  // destination register is also written on another control-flow path would have to be waited
  // for on the spot, one memory round trip per unit.
  uint32_t wi[UNITS];
#pragma unroll
  for (int k = 0; k < UNITS; ++k) {
    const uint32_t g0 = t0 + 4 * ((uint32_t)k * DS_THREADS + tid);
    const uint32_t sg = seg[k];
    const uint32_t sk = cs[sg].start_kind, src = cs[sg].src;
    const uint32_t s_begin = sk & SEG_START_MASK;
    const uint32_t s_end = sg + 1 < count ? seg_start(cs, sg + 1) : h.n_out;
    const bool is_packed = (sk & SEG_PACKED_BIT) != 0;
    const bool whole = g0 + 4 <= t1 && g0 + 4 <= s_end && !(sk & SEG_PARTIAL_BIT) && (nw <= (uint32_t)NW || !is_packed);
    // a segment holds at most PLAN_SEG_MAX_BYTES of payload: the bit offset inside it fits 32 bits
    const uint32_t byte0 = h.in_off + src;
    const uint32_t bitrel = (byte0 & 3u) * 8 + (g0 - s_begin) * h.w;
    const uint32_t kind = g0 >= t1 ? UK_NONE : (!whole ? UK_SLOW : (is_packed ? UK_PACKED : UK_RLE));
    wi[k] = kind == UK_PACKED ? (byte0 >> 2) + (bitrel >> 5) : 0u;
    raw[k].meta = (kind == UK_PACKED ? (bitrel & 31u) : 0u) | (kind << 8) | (sg << 16);
  }
  PROF(11)
#pragma unroll
  for (int k = 0; k < UNITS; ++k) {
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const uint32_t idx = wi[k] + j < h.last_word ? wi[k] + j : h.last_word;  // never past the page
      raw[k].d[j] = h.words[idx];
    }
  }
}

__device__ __forceinline__ uint32_t guarded_word(const PageHead& h, uint32_t i) {
  if (i < h.full_words) return h.words[i];
  if (i == h.full_words && h.tail_mask) return h.words[i] & h.tail_mask;
  return 0u;  // at or past the end of the stream
}

// Turns the fetched dwords into ids (values at or beyond t1 stay 0).
template <int UNITS, int NW>
__device__ __forceinline__ void extract(const PQD_LDS RleSeg* cs, uint32_t count, uint32_t t1, const PageHead& h,
                                        uint32_t tid, const UnitRaw<NW> (&raw)[UNITS], uint32_t (&id)[UNITS * 4]) {
  constexpr uint32_t t0 = 0;
  const uint32_t mask = mask32(h.w);
  bool any_slow = false;
#pragma unroll
  for (int k = 0; k < UNITS; ++k) {
    const uint32_t kind = (raw[k].meta >> 8) & 3u;
    Window5 wnd = {raw[k].d[0], NW > 1 ? raw[k].d[NW > 1 ? 1 : 0] : 0u, NW > 2 ? raw[k].d[NW > 2 ? 2 : 0] : 0u,
                   NW > 3 ? raw[k].d[NW > 3 ? 3 : 0] : 0u, NW > 4 ? raw[k].d[NW > 4 ? 4 : 0] : 0u};
    wnd.shr(raw[k].meta & 31u);
    const uint32_t a0 = wnd.d0 & mask;
    wnd.drop(h.w);
    const uint32_t a1 = wnd.d0 & mask;
    wnd.drop(h.w);
    const uint32_t a2 = wnd.d0 & mask;
    wnd.drop(h.w);
    const uint32_t a3 = wnd.d0 & mask;
    const bool packed = kind == UK_PACKED;
    const uint32_t other = kind == UK_RLE ? cs[raw[k].meta >> 16].src : 0u;  // the repeated value of an RLE run
    id[4 * k] = packed ? a0 : other;
    id[4 * k + 1] = packed ? a1 : other;
    id[4 * k + 2] = packed ? a2 : other;
    id[4 * k + 3] = packed ? a3 : other;
    any_slow |= kind == UK_SLOW;
  }
  if (any_slow) {  // rare: run boundaries that are not multiples of four, truncated final runs, wide ids
#pragma unroll 1
    for (int k = 0; k < UNITS; ++k) {
      uint32_t meta = 0;
#pragma unroll
      for (int kk = 0; kk < UNITS; ++kk) meta = kk == k ? raw[kk].meta : meta;  // k is not a compile-time constant
      if (((meta >> 8) & 3u) != UK_SLOW) continue;
      const uint32_t g0 = t0 + 4 * ((uint32_t)k * DS_THREADS + tid);
      uint32_t t = meta >> 16;
      uint32_t v[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t g = g0 + j;
        if (g >= t1) continue;
        while (t + 1 < count && seg_start(cs, t + 1) <= g) ++t;
        const uint32_t tk = cs[t].start_kind, src = cs[t].src;
        if (tk & SEG_PACKED_BIT) {
          const uint32_t byte0 = h.in_off + src;
          const uint32_t bitrel = (byte0 & 3u) * 8 + (g - (tk & SEG_START_MASK)) * h.w;
          const uint32_t wi = (byte0 >> 2) + (bitrel >> 5);
          v[j] = funnel_shr(guarded_word(h, wi), guarded_word(h, wi + 1), bitrel & 31u) & mask;
        } else {
          v[j] = src;
        }
      }
#pragma unroll
      for (int kk = 0; kk < UNITS; ++kk) {
        if (kk == k) {
          id[4 * kk] = v[0];
          id[4 * kk + 1] = v[1];
          id[4 * kk + 2] = v[2];
          id[4 * kk + 3] = v[3];
        }
      }
    }
  }
}

template <int ELEM>
__global__ __launch_bounds__(DS_THREADS) void k_dict_pages_sliced(pqd_page_result* __restrict__ results,
                                                                  const RlePagePlan* __restrict__ plans,
                                                                  const RleSeg* __restrict__ pool, uint32_t n_pages PROF_PARAM) {
  extern __shared__ u32x4 lds4[];
  uint8_t* lds = reinterpret_cast<uint8_t*>(lds4);
  DsShared* sh = reinterpret_cast<DsShared*>(lds + DS_OFF_STATE);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(PQD_LDS uint8_t*)lds;  // LDS byte address of the dynamic block

  constexpr int R = DS_R;
  constexpr int UNITS = R / 4;
  constexpr int NWP = DS_NWP;
  constexpr uint32_t S = DS_SLICE_BYTES / ELEM;  // dictionary entries per slice
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = tid >> 6, lane = tid & 63u;

  // which slice each LDS buffer holds (uniform over the workgroup)
  uint64_t tag_dict = 0;
  uint32_t tag_s0 = ~0u, tag_s1 = ~0u;
  uint32_t q = 0;  // pass counter: pass q uses buffer q & 1

  uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const uint32_t stride = gridDim.x;
  // Plan / segment buffers form a ring of three: the current page, the next one (its packed dwords
  // are fetched while the current page is gathered) and the one after (its plan is on its way).
  uint32_t pb = 0;
  fetch_plan_async(plans, pool, page, lds0 + DS_OFF_SEGS, lds0 + DS_OFF_PLANS, wave, lane);
  if (page + stride < n_pages)
    fetch_plan_async(plans, pool, page + stride, lds0 + DS_OFF_SEGS + DS_SEG_BYTES, lds0 + DS_OFF_PLANS + DS_PLAN_BYTES, wave,
                     lane);
  if (tid < 3) sh->oob_at[tid] = ~0u;
  if (tid < 4) sh->dummy[tid] = 0;
  wg_barrier();

  // a head whose loads all hit one valid dword, for when there is no page to prefetch
  PageHead idle;
  idle.words = (const PQD_GLOBAL uint32_t*)plans;
  idle.in_off = idle.last_word = idle.full_words = idle.tail_mask = 0;
  idle.n_out = idle.w = idle.flags = 0;
  idle.nseg = 1;

  UnitRaw<NWP> raw[UNITS];  // packed dwords of the page about to be decoded
  PROF_INIT
  {
    const PageHead h0 = read_head((const PQD_LDS RlePagePlan*)(lds + DS_OFF_PLANS));
    const bool ok = !(h0.flags & (PLAN_F_DONE | PLAN_F_FALLBACK)) && h0.n_out > 0;
    const PageHead& f = ok ? h0 : idle;
    locate_fetch<UNITS, NWP>((const PQD_LDS RleSeg*)(lds + DS_OFF_SEGS), f.nseg, f.n_out, f, tid, raw PROF_FN_ARG);
  }

  for (;;) {
    const uint32_t nb = pb == 2 ? 0u : pb + 1, nnb = nb == 2 ? 0u : nb + 1;
    const PQD_LDS RleSeg* cs = (const PQD_LDS RleSeg*)(lds + DS_OFF_SEGS + pb * DS_SEG_BYTES);
    const PQD_LDS RlePagePlan* lpl = (const PQD_LDS RlePagePlan*)(lds + DS_OFF_PLANS + pb * DS_PLAN_BYTES);
    const PageHead h = read_head(lpl);
    const bool skip = (h.flags & (PLAN_F_DONE | PLAN_F_FALLBACK)) != 0;
    const uint32_t n_out = skip ? 0u : h.n_out;  // at most DS_TILE_VALUES (the plan kernel sees to it)
    const uint32_t dict_len = uni(lpl->dict_len);
    const uint64_t dict_addr = ((uint64_t)uni((uint32_t)(lpl->dict >> 32)) << 32) | uni((uint32_t)lpl->dict);
    const uint64_t out_addr = ((uint64_t)uni((uint32_t)(lpl->out >> 32)) << 32) | uni((uint32_t)lpl->out);
    const int32_t page_status = (int32_t)uni((uint32_t)lpl->status);
    const uint32_t page_consumed = uni(lpl->consumed);
    const uint32_t next_page = page + stride;
    const bool has_next = next_page < n_pages;
    const uint32_t P = n_out ? (dict_len + S - 1) / S : 0u;
    const uint8_t* dict_bytes = reinterpret_cast<const uint8_t*>(dict_addr);
    if (n_out && dict_addr != tag_dict) {  // another dictionary: nothing resident is of use
      tag_dict = dict_addr;
      tag_s0 = tag_s1 = ~0u;
    }
    PROF(0)

    // ---- 1. ids into registers (their packed dwords were requested one page ago)
    uint32_t id[R];
    extract<UNITS, NWP>(cs, skip ? 1u : h.nseg, n_out, h, tid, raw, id);
    PROF(1)

    // ---- 2. prefetch: the plan of the page after the next, the packed dwords of the next page.
    // The loads are issued unconditionally (from a harmless address when there is nothing to
    // fetch) so that exactly NWP * UNITS of them are this wave's most recent memory operations
    // when it reaches the first barrier of the sweep.
    if (next_page + stride < n_pages && next_page + stride >= next_page)
      fetch_plan_async(plans, pool, next_page + stride, lds0 + DS_OFF_SEGS + nnb * DS_SEG_BYTES,
                       lds0 + DS_OFF_PLANS + nnb * DS_PLAN_BYTES, wave, lane);
    {
      const PageHead nh = read_head((const PQD_LDS RlePagePlan*)(lds + DS_OFF_PLANS + nb * DS_PLAN_BYTES));
      const bool ok = has_next && !(nh.flags & (PLAN_F_DONE | PLAN_F_FALLBACK)) && nh.n_out > 0;
      const PageHead& f = ok ? nh : idle;
      locate_fetch<UNITS, NWP>((const PQD_LDS RleSeg*)(lds + DS_OFF_SEGS + nb * DS_SEG_BYTES), f.nseg, f.n_out, f, tid, raw PROF_FN_ARG);
    }
    PROF(8)

    // ---- 3. ids beyond the dictionary: flag the first one, never look them up.
    // `done` has one bit per register slot: set = nothing (more) to look up for it.
    uint32_t done = 0;
    {
      uint32_t bad = ~0u;
#pragma unroll
      for (int k = UNITS - 1; k >= 0; --k) {
        const uint32_t g0 = 4 * ((uint32_t)k * DS_THREADS + tid);
#pragma unroll
        for (int j = 3; j >= 0; --j) {
          const bool live = g0 + j < n_out;
          const bool over = id[4 * k + j] >= dict_len;
          bad = (live && over) ? g0 + j : bad;
          done |= (!live || over) ? (1u << (4 * k + j)) : 0u;
          id[4 * k + j] = (!live || over) ? 0u : id[4 * k + j];
        }
      }
      if (bad != ~0u) atomicMin(&sh->oob_at[pb], bad);
    }
    PROF(2)

    // ---- 4. gather.  id[r] is replaced IN PLACE by the low word of its value once its slice has
    // come by (vhi[r] receives the high word of 8-byte values).
    uint32_t vhi[ELEM == 8 ? R : 1];
    if (ELEM == 8) {
#pragma unroll
      for (int r = 0; r < R; ++r) vhi[ELEM == 8 ? r : 0] = 0;
    }
    bool prefetch_only = true;  // this wave's memory operations since the last barrier: the prefetch loads alone
    for (uint32_t s = 0; s < P; ++s, ++q) {
      const uint32_t curb = q & 1u;
      // the slice this pass needs, if no earlier pass prefetched it
      if ((curb ? tag_s1 : tag_s0) != s) {
        wg_barrier();  // nobody still reads this buffer
        const uint32_t n_here = dict_len - s * S < S ? dict_len - s * S : S;
        load_slice(dict_bytes + (size_t)s * DS_SLICE_BYTES, n_here * ELEM, lds0 + curb * DS_SLICE_BYTES, wave, lane);
        if (curb)
          tag_s1 = s;
        else
          tag_s0 = s;
        prefetch_only = false;
      }
      PROF(5)
      // this pass's slice has landed; everybody is done with the other buffer
      if (prefetch_only)
        wg_barrier_keep<NWP * UNITS>();  // the prefetch loads stay in flight
      else
        wg_barrier();
      prefetch_only = false;
      PROF(3)
      // prefetch the slice of the next pass (the next page usually uses the same dictionary)
      {
        const uint32_t nx = s + 1 < P ? s + 1 : 0u;
        if ((curb ? tag_s0 : tag_s1) != nx) {
          const uint32_t n_here = dict_len - nx * S < S ? dict_len - nx * S : S;
          load_slice(dict_bytes + (size_t)nx * DS_SLICE_BYTES, n_here * ELEM, lds0 + (curb ^ 1u) * DS_SLICE_BYTES, wave, lane);
          if (curb)
            tag_s0 = nx;
          else
            tag_s1 = nx;
        }
      }
      PROF(9)
      const uint32_t s0 = s * S;
      if (ELEM == 4) {
        const PQD_LDS uint32_t* buf = (const PQD_LDS uint32_t*)(lds + curb * DS_SLICE_BYTES);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint32_t idx = id[r] - s0;
          if (idx < S && !(done & (1u << r))) {
            id[r] = buf[idx];
            done |= 1u << r;
          }
        }
      } else {
        const PQD_LDS u32x2* buf = (const PQD_LDS u32x2*)(lds + curb * DS_SLICE_BYTES);
        const PQD_LDS u32x2* dummy = (const PQD_LDS u32x2*)(sh->dummy);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint32_t idx = id[r] - s0;
          const bool hit = idx < S && !(done & (1u << r));
          const PQD_LDS u32x2* p = hit ? buf + idx : dummy;  // lanes with nothing to fetch read one shared word
          const u32x2 t = *p;
          id[r] = hit ? t.x : id[r];
          vhi[ELEM == 8 ? r : 0] = hit ? t.y : vhi[ELEM == 8 ? r : 0];
          done |= hit ? (1u << r) : 0u;
          if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // bounds the lookups in flight, and so the registers
        }
      }
      PROF(4)
    }

    // ---- 5. store, 16 bytes per lane per instruction
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
      const uint32_t g0 = 4 * ((uint32_t)k * DS_THREADS + tid);
      if (g0 >= n_out) continue;
      if (ELEM == 4) {
        PQD_GLOBAL uint32_t* o = (PQD_GLOBAL uint32_t*)out_addr + g0;
        if (g0 + 4 <= n_out) {
          u32x4_a4 r4 = {id[4 * k], id[4 * k + 1], id[4 * k + 2], id[4 * k + 3]};
          *(PQD_GLOBAL u32x4_a4*)(o) = r4;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (g0 + j < n_out) o[j] = id[4 * k + j];
        }
      } else {
        PQD_GLOBAL u32x2_a4* o = (PQD_GLOBAL u32x2_a4*)out_addr + g0;
        if (g0 + 4 <= n_out) {
          u32x4_a4 r0 = {id[4 * k], vhi[ELEM == 8 ? 4 * k : 0], id[4 * k + 1], vhi[ELEM == 8 ? 4 * k + 1 : 0]};
          u32x4_a4 r1 = {id[4 * k + 2], vhi[ELEM == 8 ? 4 * k + 2 : 0], id[4 * k + 3], vhi[ELEM == 8 ? 4 * k + 3 : 0]};
          *(PQD_GLOBAL u32x4_a4*)(o) = r0;
          *(PQD_GLOBAL u32x4_a4*)(o + 2) = r1;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (g0 + j < n_out) {
              u32x2_a4 x = {id[4 * k + j], vhi[ELEM == 8 ? 4 * k + j : 0]};
              o[j] = x;
            }
        }
      }
    }
    PROF(6)

    // ---- 6. page end.  All atomicMin on this page's oob slot are in after the barrier; the slot is
    // next used three pages on.  A page without a sweep has had no barrier that made sure the
    // plan DMA issued above has landed: it takes the full one.
    if (P == 0)
      wg_barrier();
    else
      wg_barrier_lds();
    if (!skip && tid == 0) {
      pqd_page_result r;
      const uint32_t oob = sh->oob_at[pb];
      sh->oob_at[pb] = ~0u;
      if (oob < n_out) {
        r.status = PQD_DICT_OOB;
        r.values_written = oob;
      } else {
        r.status = page_status;
        r.values_written = n_out;
      }
      r.bytes_consumed = page_consumed;
      r.reserved = 0;
      results[page] = r;
    }
    if (!has_next) break;
    pb = nb;
    page = next_page;
    PROF(7)
  }
#ifdef PQD_PROF
  if (lane == 0 && (wave == 0 || wave == 15)) {
    for (int i = 0; i < 12; ++i) prof[(blockIdx.x * 2 + (wave == 15)) * 12 + i] = pt_[i];
  }
#endif
}

static int g_cu_count = 0;
#ifdef PQD_PROF
static uint64_t* g_prof = nullptr;
static void prof_report(uint32_t grid, hipStream_t stream) {
  static const char* names[12] = {"head", "decode", "oob", "barrier", "lookup", "pre-barrier", "store", "page-end",
                                  "loads-issue", "dma-issue", "search", "classify"};
  hipStreamSynchronize(stream);
  std::vector<uint64_t> h((size_t)grid * 24);
  hipMemcpy(h.data(), g_prof, h.size() * 8, hipMemcpyDeviceToHost);
  if (!getenv("PQD_PROF_PRINT")) return;
  for (int role = 0; role < 2; ++role) {
    fprintf(stderr, "[prof] wave %d, kcycles per WG:", role ? 15 : 0);
    for (int i = 0; i < 12; ++i) {
      double sum = 0;
      for (uint32_t b = 0; b < grid; ++b) sum += (double)h[((size_t)b * 2 + role) * 12 + i];
      fprintf(stderr, " %s=%.1f", names[i], sum / grid / 1e3);
    }
    fprintf(stderr, "\n");
  }
}
#endif

hipError_t launch_dict_pages_sliced(const pqd_page_desc* descs, pqd_page_result* results, const void* plans,
                                    const void* pool, uint32_t n_pages, int elem, hipStream_t stream) {
  (void)descs;  // the plan records carry everything the kernel needs
  if (n_pages == 0) return hipSuccess;
  if (g_cu_count == 0) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return e;
    g_cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const uint32_t grid = n_pages < (uint32_t)g_cu_count ? n_pages : (uint32_t)g_cu_count;
  const RlePagePlan* pp = reinterpret_cast<const RlePagePlan*>(plans);
  const RleSeg* sp = reinterpret_cast<const RleSeg*>(pool);
  hipError_t e;
#ifdef PQD_PROF
  if (!g_prof && (e = hipMalloc((void**)&g_prof, 1024 * 24 * 8)) != hipSuccess) return e;
#endif
  if (elem == 4) {
    auto k = k_dict_pages_sliced<4>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, DS_LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(DS_THREADS), DS_LDS_BYTES, stream, results, pp, sp, n_pages PROF_ARG);
  } else {
    auto k = k_dict_pages_sliced<8>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, DS_LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(DS_THREADS), DS_LDS_BYTES, stream, results, pp, sp, n_pages PROF_ARG);
  }
#ifdef PQD_PROF
  prof_report(grid, stream);
#endif
  return hipGetLastError();
}

}  // namespace pqd
