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
//   - 14 of the 16 waves hold values: a tile of up to 896*R ids is unpacked into REGISTERS from
//     packed dwords read straight from global memory (coalesced; each lane owns units of four
//     consecutive values);
//   - the other 2 waves are loaders: they stream the dictionary through LDS in 64 KiB slices,
//     double-buffered, by LDS-DMA (global_load_lds_dwordx4).  Only the loaders wait for their own
//     memory operations at the workgroup barriers, so the value waves' loads and stores stay in
//     flight across barriers;
//   - each pass resolves the ids that fall into the resident slice; a resolved id is replaced in
//     place by its value;
//   - while one page is being gathered, the next page's plan, segments and packed dwords are
//     already being fetched, so no memory round trip is exposed between pages;
//   - the values are stored once, 16 bytes per lane, fully coalesced.
// HBM traffic per value: w/8 bytes of ids + ELEM bytes out; the slices are L2 traffic
// (dictionary bytes per tile, i.e. a few bytes per value).  Small dictionaries (one slice) stay
// resident across pages.  Dictionaries too large to sweep fall back to a direct gather.
#include "pqd_device.hpp"
#include "pqd_kernels.hpp"
#include "pqd_rle_plan.hpp"

namespace pqd {

constexpr int DS_THREADS = 1024;
constexpr int DS_LOADER_WAVES = 2;
constexpr int DS_VALUE_THREADS = DS_THREADS - 64 * DS_LOADER_WAVES;  // 896
constexpr int DS_SLICE_BYTES = 64 * 1024;
constexpr int DS_MAX_PASSES = 32;
constexpr int DS_SEG_BYTES = SEG_CHUNK * 8;  // one segment chunk, 1 KiB

constexpr int DS_OFF_SEGS = 2 * DS_SLICE_BYTES;               // two segment-chunk buffers
constexpr int DS_OFF_PLANS = DS_OFF_SEGS + 2 * DS_SEG_BYTES;  // two plan records
constexpr int DS_OFF_STATE = DS_OFF_PLANS + 2 * 128;
constexpr int DS_LDS_BYTES = DS_OFF_STATE + 64;
static_assert(DS_LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(DS_OFF_STATE % 16 == 0, "state block alignment");

#define PQD_GLOBAL __attribute__((address_space(1)))
#define PQD_LDS __attribute__((address_space(3)))
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

struct DsShared {
  uint32_t oob_at;
  uint32_t pad[3];
  uint32_t dummy[4];  // what lanes without a lookup read (never written after the prologue)
};

// Workgroup barrier that does not drain the value waves' global loads and stores: LDS traffic is
// waited for by everybody, vector-memory traffic (the LDS-DMA) only by the loader waves.
__device__ __forceinline__ void wg_barrier(bool loader) {
  if (loader)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Loader waves: fill one slice buffer by LDS-DMA; a wave instruction moves 64 x 16 bytes to
// consecutive LDS addresses.  `bytes` is the valid length of the slice.
__device__ __forceinline__ void load_slice(const uint8_t* slice_src, uint32_t bytes, uint8_t* buf, uint32_t lwave,
                                           uint32_t lane) {
  const uint32_t pieces = (bytes + 15) >> 4;
#pragma unroll 4
  for (int i = 0; i < DS_SLICE_BYTES / 16 / (64 * DS_LOADER_WAVES); ++i) {
    const uint32_t piece0 = ((uint32_t)i * DS_LOADER_WAVES + lwave) * 64;  // wave-uniform
    if (piece0 >= pieces) break;
    if (piece0 + lane < pieces)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(slice_src + (size_t)(piece0 + lane) * 16),
                                       (lds_void_t*)(buf + (size_t)piece0 * 16), 16, 0, 0);
  }
}

// Loader waves: LDS-DMA of a page's plan record (128 B) and first segment chunk (1 KiB; chunk 0
// of page p is pool chunk p).
__device__ __forceinline__ void fetch_plan_async(const RlePagePlan* plans, const RleSeg* pool, uint32_t page,
                                                 uint8_t* seg_buf, uint8_t* plan_buf, uint32_t lwave, uint32_t lane) {
  if (lwave == 0) {
    const uint8_t* src = reinterpret_cast<const uint8_t*>(pool + (size_t)page * SEG_CHUNK);
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + (size_t)lane * 16), (lds_void_t*)seg_buf, 16, 0, 0);
  } else if (lwave == 1) {
    if (lane < 8) {
      const uint8_t* src = reinterpret_cast<const uint8_t*>(plans + page);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + (size_t)lane * 16), (lds_void_t*)plan_buf, 16, 0, 0);
    }
  }
}

__device__ __forceinline__ uint32_t uni(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }

// Page constants, read from the plan record in LDS and held in scalar registers.
struct PageHead {
  const PQD_GLOBAL uint32_t* words;  // `data` rounded down to a dword
  uint32_t in_off;                   // data & 3
  uint32_t last_word;                // index of the last dword that holds a byte of the page
  uint32_t full_words;               // dwords lying wholly before the end of the stream
  uint32_t tail_mask;                // valid bytes of dword `full_words` (0: none)
  uint32_t n_out, w, nseg, flags;
};

__device__ __forceinline__ PageHead read_head(const uint8_t* plan_buf) {
  const PQD_LDS RlePagePlan* pl = (const PQD_LDS RlePagePlan*)(plan_buf);
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
  uint32_t meta;  // bits 0..4: bit offset inside d[0]; bits 8..9: kind
};
constexpr uint32_t UK_NONE = 0, UK_PACKED = 1, UK_RLE = 2, UK_SLOW = 3;

__device__ __forceinline__ uint32_t seg_start(const PQD_LDS RleSeg* cs, uint32_t i) {
  return cs[i].start_kind & SEG_START_MASK;
}

// For every unit of the tile starting at t0 that lies in the chunk's value range [lo, hi): find
// its segment (binary search in LDS) and ISSUE the global loads of its packed dwords.  Nothing
// waits on the loads here, so all units' loads are in flight together (and, when this is called
// for the next page during a sweep, they complete behind the sweep).  NW = dwords kept per unit:
// units of wider ids (ceil((31 + 4w) / 32) > NW) are marked for the value-by-value path.
template <int UNITS, int NW>
__device__ __forceinline__ void locate_fetch(const PQD_LDS RleSeg* cs, uint32_t count, uint32_t chunk_end,
                                             uint32_t t0, uint32_t lo, uint32_t hi, const PageHead& h, uint32_t vt,
                                             UnitRaw<NW> (&raw)[UNITS]) {
  const uint32_t nw = (62u + 4u * h.w) >> 5;
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < UNITS; ++k) {
    raw[k].meta = UK_NONE << 8;
#pragma unroll
    for (int j = 0; j < NW; ++j) raw[k].d[j] = 0;
    const uint32_t g0 = t0 + 4 * ((uint32_t)k * DS_VALUE_THREADS + vt);
    if (g0 + 4 <= lo || g0 >= hi) continue;
    const uint32_t l0 = g0 > lo ? g0 : lo;
    uint32_t a = s, b = count;  // start[a] <= l0 < start[b]
    while (b - a > 1) {
      const uint32_t m = (a + b) >> 1;
      if (seg_start(cs, m) <= l0)
        a = m;
      else
        b = m;
    }
    s = a;
    const uint32_t sk = cs[s].start_kind;
    const uint32_t s_begin = sk & SEG_START_MASK;
    const uint32_t s_end = s + 1 < count ? seg_start(cs, s + 1) : chunk_end;
    // widths whose units need more than NW dwords take the value-by-value path
    if (g0 >= lo && g0 + 4 <= hi && g0 + 4 <= s_end && !(sk & SEG_PARTIAL_BIT) && (nw <= (uint32_t)NW || !(sk & SEG_PACKED_BIT))) {
      const uint32_t src = cs[s].src;
      if (sk & SEG_PACKED_BIT) {
        // a segment holds at most PLAN_SEG_MAX_BYTES of payload: the bit offset inside it fits 32 bits
        const uint32_t byte0 = h.in_off + src;
        const uint32_t bitrel = (byte0 & 3u) * 8 + (g0 - s_begin) * h.w;
        const uint32_t wi = (byte0 >> 2) + (bitrel >> 5);
        raw[k].meta = (bitrel & 31u) | (UK_PACKED << 8);
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          if ((uint32_t)j < nw) {  // uniform
            const uint32_t idx = wi + j < h.last_word ? wi + j : h.last_word;  // never past the page
            raw[k].d[j] = h.words[idx];
          }
        }
      } else {
        raw[k].meta = UK_RLE << 8;
        raw[k].d[0] = src;
      }
    } else {
      raw[k].meta = UK_SLOW << 8;
      raw[k].d[0] = s;
    }
  }
}

__device__ __forceinline__ uint32_t guarded_word(const PageHead& h, uint32_t i) {
  if (i < h.full_words) return h.words[i];
  if (i == h.full_words && h.tail_mask) return h.words[i] & h.tail_mask;
  return 0u;  // at or past the end of the stream
}

// Turns the fetched dwords into ids.  Units that straddle segments (or sit in a segment cut short
// by the end of the stream) are done value by value with loads of their own.
template <int UNITS, int NW>
__device__ __forceinline__ void extract(const PQD_LDS RleSeg* cs, uint32_t count, uint32_t t0, uint32_t lo,
                                        uint32_t hi, const PageHead& h, uint32_t vt, const UnitRaw<NW> (&raw)[UNITS],
                                        uint32_t (&id)[UNITS * 4]) {
  const uint32_t mask = mask32(h.w);
#pragma unroll
  for (int k = 0; k < UNITS; ++k) {
    const uint32_t kind = raw[k].meta >> 8;
    if (kind == UK_PACKED) {
      Window5 wnd = {raw[k].d[0], NW > 1 ? raw[k].d[NW > 1 ? 1 : 0] : 0u, NW > 2 ? raw[k].d[NW > 2 ? 2 : 0] : 0u,
                     NW > 3 ? raw[k].d[NW > 3 ? 3 : 0] : 0u, NW > 4 ? raw[k].d[NW > 4 ? 4 : 0] : 0u};
      wnd.shr(raw[k].meta & 31u);
      id[4 * k] = wnd.d0 & mask;
      wnd.drop(h.w);
      id[4 * k + 1] = wnd.d0 & mask;
      wnd.drop(h.w);
      id[4 * k + 2] = wnd.d0 & mask;
      wnd.drop(h.w);
      id[4 * k + 3] = wnd.d0 & mask;
    } else if (kind == UK_RLE) {
      id[4 * k] = id[4 * k + 1] = id[4 * k + 2] = id[4 * k + 3] = raw[k].d[0];
    } else if (kind == UK_SLOW) {
      const uint32_t g0 = t0 + 4 * ((uint32_t)k * DS_VALUE_THREADS + vt);
      uint32_t t = raw[k].d[0];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t g = g0 + j;
        if (g < lo || g >= hi) continue;
        while (t + 1 < count && seg_start(cs, t + 1) <= g) ++t;
        const uint32_t tk = cs[t].start_kind, src = cs[t].src;
        if (tk & SEG_PACKED_BIT) {
          const uint32_t byte0 = h.in_off + src;
          const uint32_t bitrel = (byte0 & 3u) * 8 + (g - (tk & SEG_START_MASK)) * h.w;
          const uint32_t wi = (byte0 >> 2) + (bitrel >> 5);
          id[4 * k + j] = funnel_shr(guarded_word(h, wi), guarded_word(h, wi + 1), bitrel & 31u) & mask;
        } else {
          id[4 * k + j] = src;
        }
      }
    }
  }
}

template <int ELEM, int R>
__global__ __launch_bounds__(DS_THREADS) void k_dict_pages_sliced(pqd_page_result* __restrict__ results,
                                                                  const RlePagePlan* __restrict__ plans,
                                                                  const RleSeg* __restrict__ pool, uint32_t n_pages) {
  extern __shared__ u32x4 lds4[];
  uint8_t* lds = reinterpret_cast<uint8_t*>(lds4);
  DsShared* sh = reinterpret_cast<DsShared*>(lds + DS_OFF_STATE);

  constexpr int UNITS = R / 4;
  constexpr int NWP = 3;  // dwords per unit kept for the prefetched page (enough for widths <= 16)
  constexpr uint32_t TILE = DS_VALUE_THREADS * R;
  constexpr uint32_t S = DS_SLICE_BYTES / ELEM;  // dictionary entries per slice
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = tid >> 6, lane = tid & 63u;
  const bool loader = wave >= (uint32_t)(DS_THREADS / 64 - DS_LOADER_WAVES);  // wave-uniform role
  const uint32_t lwave = wave - (DS_THREADS / 64 - DS_LOADER_WAVES);
  const uint32_t vt = tid;  // value-thread index (only meaningful for value waves)

  // which slice each LDS buffer holds (uniform over the workgroup; both roles track it)
  uint64_t tag_dict0 = 0, tag_dict1 = 0;
  uint32_t tag_s0 = ~0u, tag_s1 = ~0u;
  uint32_t q = 0;  // pass counter: pass q uses buffer q & 1

  uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  uint32_t pb = 0;  // which plan / segment buffer holds the current page
  if (loader) fetch_plan_async(plans, pool, page, lds + DS_OFF_SEGS, lds + DS_OFF_PLANS, lwave, lane);
  if (tid == 0) sh->oob_at = ~0u;
  if (tid < 4) sh->dummy[tid] = 0;
  wg_barrier(loader);

  UnitRaw<NWP> raw[UNITS];
  bool have_raw = false;  // raw[] already holds the first tile of the current page

  for (;;) {
    const uint8_t* plan_buf = lds + DS_OFF_PLANS + pb * 128;
    const PQD_LDS RleSeg* cs = (const PQD_LDS RleSeg*)(lds + DS_OFF_SEGS + pb * DS_SEG_BYTES);
    const PQD_LDS RlePagePlan* lpl = (const PQD_LDS RlePagePlan*)(plan_buf);
    const PageHead h = read_head(plan_buf);
    const bool skip = (h.flags & (PLAN_F_DONE | PLAN_F_FALLBACK)) != 0;
    const uint32_t n_out = skip ? 0u : h.n_out, nseg = h.nseg;
    const uint32_t dict_len = uni(lpl->dict_len);
    const uint64_t dict_addr = ((uint64_t)uni((uint32_t)(lpl->dict >> 32)) << 32) | uni((uint32_t)lpl->dict);
    const uint64_t out_addr = ((uint64_t)uni((uint32_t)(lpl->out >> 32)) << 32) | uni((uint32_t)lpl->out);
    const int32_t page_status = (int32_t)uni((uint32_t)lpl->status);
    const uint32_t page_consumed = uni(lpl->consumed);
    const uint32_t next_page = page + gridDim.x;
    const bool has_next = next_page < n_pages;
    bool next_fetched = false, next_raw = false;

    const uint32_t P = (dict_len + S - 1) / S;
    // chunk 0 is in LDS; later chunks replace it in the same buffer (the cursor only moves forward)
    uint32_t chunk_i = 0, loaded_chunk = 0;
    uint32_t count = nseg < (uint32_t)SEG_CHUNK ? nseg : (uint32_t)SEG_CHUNK;
    uint32_t chunk_end = n_out;
    if (nseg > (uint32_t)SEG_CHUNK && !skip)
      chunk_end = uni(((const PQD_GLOBAL RleSeg*)pool)[(size_t)lpl->chunk[1] * SEG_CHUNK].start_kind) & SEG_START_MASK;

    for (uint32_t t0 = 0; t0 < n_out; t0 += TILE) {
      const uint32_t t1 = (n_out - t0 > TILE) ? t0 + TILE : n_out;
      const bool last_tile = t1 == n_out;
      uint32_t id[R];
#pragma unroll
      for (int r = 0; r < R; ++r) id[r] = 0;

      // ---- 1. ids of [t0, t1) into registers, one segment chunk at a time
      for (;;) {
        if (loaded_chunk != chunk_i) {
          wg_barrier(loader);  // earlier readers of the segment buffer are done
          const uint32_t first = chunk_i * SEG_CHUNK;
          count = nseg - first < (uint32_t)SEG_CHUNK ? nseg - first : (uint32_t)SEG_CHUNK;
          const PQD_GLOBAL RleSeg* src = (const PQD_GLOBAL RleSeg*)pool + (size_t)lpl->chunk[chunk_i] * SEG_CHUNK;
          PQD_LDS RleSeg* dst = (PQD_LDS RleSeg*)(lds + DS_OFF_SEGS + pb * DS_SEG_BYTES);
          if (tid < count) *(PQD_LDS uint64_t*)(dst + tid) = *(const PQD_GLOBAL uint64_t*)(src + tid);
          chunk_end = n_out;
          if (first + count < nseg)
            chunk_end = uni(((const PQD_GLOBAL RleSeg*)pool)[(size_t)lpl->chunk[chunk_i + 1] * SEG_CHUNK].start_kind) &
                        SEG_START_MASK;
          loaded_chunk = chunk_i;
          have_raw = false;
          wg_barrier(loader);
        }
        const uint32_t cv0 = seg_start(cs, 0);
        const uint32_t lo = cv0 > t0 ? cv0 : t0, hi = chunk_end < t1 ? chunk_end : t1;
        if (lo < hi && !loader) {
          if (!have_raw) locate_fetch<UNITS, NWP>(cs, count, chunk_end, t0, lo, hi, h, vt, raw);
          extract<UNITS, NWP>(cs, count, t0, lo, hi, h, vt, raw, id);
        }
        have_raw = false;
        if (chunk_end >= t1) break;
        ++chunk_i;
      }

      // ---- 2. ids beyond the dictionary: flag the first one, never look them up.
      // `done` has one bit per register slot: set = nothing (more) to look up for it.
      uint32_t done = 0;
      if (!loader) {
        uint32_t bad = ~0u;
#pragma unroll
        for (int k = UNITS - 1; k >= 0; --k) {
          const uint32_t g0 = t0 + 4 * ((uint32_t)k * DS_VALUE_THREADS + vt);
#pragma unroll
          for (int j = 3; j >= 0; --j) {
            const bool live = g0 + j < t1;
            if (live && id[4 * k + j] >= dict_len) bad = g0 + j;
            if (!live || id[4 * k + j] >= dict_len) {
              done |= 1u << (4 * k + j);
              id[4 * k + j] = 0;
            }
          }
        }
        if (bad != ~0u) atomicMin(&sh->oob_at, bad);
      }

      // ---- 3. gather.  id[r] is replaced IN PLACE by the low word of its value once its slice has
      // come by (vhi[r] receives the high word of 8-byte values).  The lookups of one pass are
      // branch-free so that they overlap: lanes with nothing to fetch read one shared dummy word
      // (a broadcast, no bank conflict).
      uint32_t vhi[ELEM == 8 ? R : 1];
      if (ELEM == 8) {
#pragma unroll
        for (int r = 0; r < R; ++r) vhi[ELEM == 8 ? r : 0] = 0;
      }
      const bool sweep = P > 0 && (dict_addr & 15u) == 0 &&
                         (P == 1 || (P <= (uint32_t)DS_MAX_PASSES && (uint64_t)dict_len * ELEM <= 32ull * (t1 - t0)));
      if (sweep) {
        const uint8_t* dict_bytes = reinterpret_cast<const uint8_t*>(dict_addr);
        for (uint32_t s = 0; s < P; ++s, ++q) {
          const uint32_t curb = q & 1u;
          // the slice this pass needs, if no earlier pass prefetched it
          {
            const uint64_t td = curb ? tag_dict1 : tag_dict0;
            const uint32_t ts = curb ? tag_s1 : tag_s0;
            if (td != dict_addr || ts != s) {
              wg_barrier(loader);  // nobody still reads this buffer
              const uint32_t n_here = dict_len - s * S < S ? dict_len - s * S : S;
              if (loader)
                load_slice(dict_bytes + (size_t)s * DS_SLICE_BYTES, n_here * ELEM, lds + curb * DS_SLICE_BYTES, lwave, lane);
              if (curb) {
                tag_dict1 = dict_addr;
                tag_s1 = s;
              } else {
                tag_dict0 = dict_addr;
                tag_s0 = s;
              }
            }
          }
          wg_barrier(loader);  // the loaders' DMA has landed; everybody is done with the other buffer
          // prefetch the slice of the next pass (the next tile / page usually uses the same dictionary)
          {
            const uint32_t nx = s + 1 < P ? s + 1 : 0u;
            const uint64_t td = curb ? tag_dict0 : tag_dict1;
            const uint32_t ts = curb ? tag_s0 : tag_s1;
            if (td != dict_addr || ts != nx) {
              const uint32_t n_here = dict_len - nx * S < S ? dict_len - nx * S : S;
              if (loader)
                load_slice(dict_bytes + (size_t)nx * DS_SLICE_BYTES, n_here * ELEM, lds + (curb ^ 1u) * DS_SLICE_BYTES, lwave,
                           lane);
              if (curb) {
                tag_dict0 = dict_addr;
                tag_s0 = nx;
              } else {
                tag_dict1 = dict_addr;
                tag_s1 = nx;
              }
            }
          }
          // the next page rides behind this page's last sweep: its plan and segments are fetched
          // during pass 0 (loaders), its packed dwords during pass 1 (value waves)
          if (last_tile && has_next) {
            if (s == 0) {
              if (loader)
                fetch_plan_async(plans, pool, next_page, lds + DS_OFF_SEGS + (pb ^ 1u) * DS_SEG_BYTES,
                                 lds + DS_OFF_PLANS + (pb ^ 1u) * 128, lwave, lane);
              next_fetched = true;
            } else if (s == 1) {
              const uint8_t* nplan = lds + DS_OFF_PLANS + (pb ^ 1u) * 128;
              const PageHead nh = read_head(nplan);
              if (!(nh.flags & (PLAN_F_DONE | PLAN_F_FALLBACK)) && nh.nseg <= (uint32_t)SEG_CHUNK && nh.n_out > 0) {
                if (!loader) {
                  const PQD_LDS RleSeg* ncs = (const PQD_LDS RleSeg*)(lds + DS_OFF_SEGS + (pb ^ 1u) * DS_SEG_BYTES);
                  const uint32_t nhi = nh.n_out < TILE ? nh.n_out : TILE;
                  locate_fetch<UNITS, NWP>(ncs, nh.nseg, nh.n_out, 0u, 0u, nhi, nh, vt, raw);
                }
                next_raw = true;
              }
            }
          }
          if (loader) continue;
          const uint32_t s0 = s * S;
          if (ELEM == 4) {
            const PQD_LDS uint32_t* buf = (const PQD_LDS uint32_t*)(lds + curb * DS_SLICE_BYTES);
            const PQD_LDS uint32_t* dummy = (const PQD_LDS uint32_t*)(sh->dummy);
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const uint32_t idx = id[r] - s0;
              const bool hit = idx < S && !(done & (1u << r));
              const PQD_LDS uint32_t* p = hit ? buf + idx : dummy;
              const uint32_t t = *p;
              id[r] = hit ? t : id[r];
              done |= hit ? (1u << r) : 0u;
              if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // at most 8 lookups in flight: bounds the registers
            }
          } else {
            const PQD_LDS u32x2* buf = (const PQD_LDS u32x2*)(lds + curb * DS_SLICE_BYTES);
            const PQD_LDS u32x2* dummy = (const PQD_LDS u32x2*)(sh->dummy);
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const uint32_t idx = id[r] - s0;
              const bool hit = idx < S && !(done & (1u << r));
              const PQD_LDS u32x2* p = hit ? buf + idx : dummy;
              const u32x2 t = *p;
              id[r] = hit ? t.x : id[r];
              vhi[ELEM == 8 ? r : 0] = hit ? t.y : vhi[ELEM == 8 ? r : 0];
              done |= hit ? (1u << r) : 0u;
              if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // at most 4 lookups in flight: bounds the registers
            }
          }
        }
      } else if (!loader) {
        // direct gather through the vector cache (huge dictionaries, or too few values to pay for a sweep)
        if (ELEM == 4) {
          const PQD_GLOBAL uint32_t* dict = (const PQD_GLOBAL uint32_t*)dict_addr;
#pragma unroll
          for (int r = 0; r < R; ++r)
            if (!(done & (1u << r))) id[r] = dict[id[r]];
        } else {
          const PQD_GLOBAL u32x2_a4* dict = (const PQD_GLOBAL u32x2_a4*)dict_addr;
#pragma unroll
          for (int r = 0; r < R; ++r)
            if (!(done & (1u << r))) {
              const u32x2_a4 x = dict[id[r]];
              id[r] = x.x;
              vhi[ELEM == 8 ? r : 0] = x.y;
            }
        }
      }

      // ---- 4. store, 16 bytes per lane per instruction
      if (!loader) {
#pragma unroll
        for (int k = 0; k < UNITS; ++k) {
          const uint32_t g0 = t0 + 4 * ((uint32_t)k * DS_VALUE_THREADS + vt);
          if (g0 >= t1) continue;
          if (ELEM == 4) {
            PQD_GLOBAL uint32_t* o = (PQD_GLOBAL uint32_t*)out_addr + g0;
            if (g0 + 4 <= t1) {
              u32x4_a4 r4 = {id[4 * k], id[4 * k + 1], id[4 * k + 2], id[4 * k + 3]};
              *(PQD_GLOBAL u32x4_a4*)(o) = r4;
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (g0 + j < t1) o[j] = id[4 * k + j];
            }
          } else {
            PQD_GLOBAL u32x2_a4* o = (PQD_GLOBAL u32x2_a4*)out_addr + g0;
            if (g0 + 4 <= t1) {
              u32x4_a4 r0 = {id[4 * k], vhi[ELEM == 8 ? 4 * k : 0], id[4 * k + 1], vhi[ELEM == 8 ? 4 * k + 1 : 0]};
              u32x4_a4 r1 = {id[4 * k + 2], vhi[ELEM == 8 ? 4 * k + 2 : 0], id[4 * k + 3], vhi[ELEM == 8 ? 4 * k + 3 : 0]};
              *(PQD_GLOBAL u32x4_a4*)(o) = r0;
              *(PQD_GLOBAL u32x4_a4*)(o + 2) = r1;
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (g0 + j < t1) {
                  u32x2_a4 x = {id[4 * k + j], vhi[ELEM == 8 ? 4 * k + j : 0]};
                  o[j] = x;
                }
            }
          }
        }
      }
    }

    wg_barrier(loader);  // all atomicMin on oob_at are in; everybody is done with this page's buffers
    if (!skip && tid == 0) {
      pqd_page_result r;
      const uint32_t oob = sh->oob_at;
      sh->oob_at = ~0u;  // for the next page; the barrier below orders it before any new atomicMin
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
    if (!next_fetched && loader)
      fetch_plan_async(plans, pool, next_page, lds + DS_OFF_SEGS + (pb ^ 1u) * DS_SEG_BYTES,
                       lds + DS_OFF_PLANS + (pb ^ 1u) * 128, lwave, lane);
    wg_barrier(loader);  // the next page's plan and segments are in LDS; oob_at has been read
    pb ^= 1u;
    page = next_page;
    have_raw = next_raw;
  }
}

static int g_cu_count = 0;

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
  if (elem == 4) {
    auto k = k_dict_pages_sliced<4, 24>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, DS_LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(DS_THREADS), DS_LDS_BYTES, stream, results, pp, sp, n_pages);
  } else {
    auto k = k_dict_pages_sliced<8, 24>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, DS_LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(DS_THREADS), DS_LDS_BYTES, stream, results, pp, sp, n_pages);
  }
  return hipGetLastError();
}

}  // namespace pqd
