// Dictionary-encoded pages, expansion + gather, for dictionaries larger than the L1 cache:
// the dictionary is streamed through LDS in slices and every lookup is an LDS read.
//
// Replaces the read loop of DictionaryValuesReader (parquet-column/.../values/dictionary/
// DictionaryValuesReader.java:75-118: value = dictionary.decodeToX(decoder.readInt())) for whole
// pages.  A random gather from a 256-512 KiB dictionary through the vector cache costs a cache
// line per value and runs at about 0.3 values/clk/CU; an LDS gather runs an order of magnitude
// faster, so instead of sending each id to the dictionary the dictionary is brought to the ids:
//
//   - one 1024-thread workgroup per CU walks over CHUNKS of the run plan that k_rle_plan_chunks
//     made (up to 20480 consecutive values of a page, at most 231 run segments), so nothing here
//     is serial and pages of any length or run count are handled;
//   - a tile of up to 1024*R ids is unpacked into REGISTERS from packed dwords read straight from
//     global memory (coalesced; each lane owns units of four consecutive values);
//   - the dictionary streams through LDS in 64 KiB slices, double-buffered, by LDS-DMA
//     (global_load_lds_dwordx4) issued by all 16 waves;
//   - each pass looks up the ids that fall into the resident slice (their values go to registers
//     of their own; an id is found by exactly one pass).  4-byte dictionaries are swept as they
//     are; 8-byte dictionaries are swept as two dword PLANES (a copy made once per call by
//     dict_plane_piece: all low dwords, then all high dwords), a quarter of a plane (16384 entries,
//     64 KiB) per pass, so that an id's LDS address is the same in every pass and a lookup is one
//     byte-selecting compare that puts the quarter's lanes into EXEC plus one 4-byte LDS read;
//   - while one page is being gathered, the next page's plan, segments and packed dwords are
//     already being fetched (behind passes 0 and 1), so no memory round trip is exposed between
//     pages;
//   - the values are stored once, 16 bytes per lane, with the nontemporal hint (they are not read
//     again, and left in L2 they push out the dictionary slices).
// HBM traffic per value: w/8 bytes of ids + ELEM bytes out; the slices are L2 traffic
// (dictionary bytes per tile, i.e. a few bytes per value).  Small dictionaries (one slice) stay
// resident across chunks.  Pages this kernel is not made for (ids wider than 16 bits, a dictionary
// too large to sweep) are flagged by the plan kernel and left to the generic kernel.
#include <hip/hip_ext.h>

#include "pqd_device.hpp"
#include <cstdlib>
#include <cstring>
#ifdef PQD_PROF
#include <cstdio>
#include <vector>
#endif
#include "pqd_kernels.hpp"
#include "pqd_rle_plan.hpp"

namespace pqd {

constexpr int DS_THREADS = 1024;
constexpr int DS_WAVES = DS_THREADS / 64;
constexpr int DS_R = DS_TILE_VALUES / DS_THREADS;  // values per lane per tile
constexpr int DS_SEG_BYTES = SEG_CHUNK * 8;         // one chunk of the plan: segments, header, block table

constexpr int DS_OFF_SEGS = 2 * DS_SLICE_BYTES;  // three chunk buffers
constexpr int DS_OFF_STATE = DS_OFF_SEGS + 3 * DS_SEG_BYTES;
// per wave: a list of up to 64 units left to the value-by-value pass (4 bytes each) and their ids (16 bytes each)
constexpr int DS_SLOW_UNITS = 64;
constexpr int DS_SLOW_BYTES = DS_SLOW_UNITS * (4 + 16);
constexpr int DS_OFF_SLOW = DS_OFF_STATE + 64;
constexpr int DS_LDS_BYTES = DS_OFF_SLOW + DS_WAVES * DS_SLOW_BYTES;
static_assert(DS_LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(DS_OFF_STATE % 16 == 0, "state block alignment");
static_assert(DS_R % 4 == 0 && DS_R <= 32, "a lane holds whole units, one flag bit per value");


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
  uint32_t oob_at[3];  // first out-of-range id (output index) of the chunk in buffer 0 / 1 / 2
  uint32_t pad;
};

__device__ __forceinline__ uint32_t uni(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }

// LDS-DMA of 64 x 16 bytes: lane l's 16 bytes at gbase + goff land at LDS byte address
// lds_base + 16*l (gbase and lds_base wave-uniform).  Issued through inline assembly on purpose:
// the compiler orders every LDS-DMA it knows of against all later LDS accesses and LDS-DMAs with a
// full s_waitcnt vmcnt(0), which serialises the DMA instructions of a wave and stalls the lookups
// behind the slice being prefetched into the OTHER buffer.  Completion is awaited explicitly in
// wg_barrier().
__device__ __forceinline__ void dma16(const void* gbase, uint32_t goff, uint32_t lds_base) {
  // the base must sit in scalar registers whatever the compiler thinks of its uniformity
  const uint64_t g = (uint64_t)(uintptr_t)gbase;
  const uint64_t gs = ((uint64_t)uni((uint32_t)(g >> 32)) << 32) | uni((uint32_t)g);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base), "v"(goff), "s"(gs)
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
  if (bytes == DS_SLICE_BYTES) {  // no per-lane bounds to check
#pragma unroll
    for (int i = 0; i < (int)(DS_SLICE_BYTES / 16 / DS_THREADS); ++i) {
      const uint32_t piece0 = ((uint32_t)i * DS_WAVES + wave) * 64;  // wave-uniform
      dma16(slice_src + (size_t)piece0 * 16, lane * 16, lds_buf + piece0 * 16);
    }
    return;
  }
  const uint32_t pieces = (bytes + 15) >> 4;
#pragma unroll
  for (int i = 0; i < (int)(DS_SLICE_BYTES / 16 / DS_THREADS); ++i) {
    const uint32_t piece0 = ((uint32_t)i * DS_WAVES + wave) * 64;  // wave-uniform
    // (the lane offset is the same register for every piece: per-piece offsets would be hoisted out
    // of the chunk loop and held in registers the kernel does not have)
    if (piece0 < pieces && lane < pieces - piece0) dma16(slice_src + (size_t)piece0 * 16, lane * 16, uni(lds_buf + piece0 * 16));
  }
}

// One chunk of the plan (2 KiB: 1 KiB per wave, waves 0 and 1) into the LDS buffer at byte address `buf`.
__device__ __forceinline__ void fetch_chunk_async(const RleSeg* pool, uint32_t chunk, uint32_t buf, uint32_t wave,
                                                  uint32_t lane) {
  if (wave < (uint32_t)(DS_SEG_BYTES / 1024))
    dma16(pool + (size_t)chunk * SEG_CHUNK, wave * 1024 + lane * 16, uni(buf + wave * 1024));
}

// Chunk constants, read from the chunk header in LDS and held in scalar registers.
struct PageHead {
  const PQD_GLOBAL uint32_t* words;  // `data` rounded down to a dword
  const PQD_GLOBAL uint32_t* load_words;  // base of the 12-byte unit loads (`words`, or a harmless address)
  uint32_t in_off;                   // data & 3
  uint32_t max_wi;                   // largest dword index a 12-byte unit load may start at
  uint32_t full_words;               // dwords lying wholly before the end of the stream
  uint32_t tail_mask;                // valid bytes of dword `full_words` (0: none)
  uint32_t v0, v1, w;                // output values [v0, v1) of the page
  bool no_fast;                      // no unit of this chunk can take the 12-byte path
};

// `spare` is any 12 readable bytes: what the unit loads read when there is nothing to read.
// `valid` false: the buffer holds no chunk (none left); the head then describes an empty range.
__device__ __forceinline__ PageHead read_head(const PQD_LDS RleSeg* chunk_buf, bool valid, const void* spare) {
  const PQD_LDS RleChunkHead* hd = (const PQD_LDS RleChunkHead*)(chunk_buf + SEG_HEAD_SLOT);
  PageHead h;
  const uint32_t dlo = uni((uint32_t)hd->data), dhi = uni((uint32_t)(hd->data >> 32));
  const uint64_t data = ((uint64_t)dhi << 32) | dlo;
  h.in_off = dlo & 3u;
  h.words = (const PQD_GLOBAL uint32_t*)(data - h.in_off);
  const uint32_t data_len = uni(hd->data_len), stream_end = uni(hd->stream_end);
  const uint32_t last_word = data_len ? (h.in_off + data_len - 1) >> 2 : 0u;  // last dword holding a byte of the page
  const uint32_t end_bytes = h.in_off + stream_end;
  h.full_words = end_bytes >> 2;
  h.tail_mask = (end_bytes & 3u) ? ((1u << (8 * (end_bytes & 3u))) - 1u) : 0u;
  h.v0 = valid ? uni(hd->v0) : 0u;
  h.v1 = valid ? uni(hd->v1) : 0u;
  h.w = uni(hd->w);
  // ids of up to 16 bits: four of them, at any bit offset, lie inside three dwords
  h.no_fast = !valid || last_word < 2 || h.w > 16;
  h.max_wi = h.no_fast ? 0u : last_word - 2;
  h.load_words = h.no_fast ? (const PQD_GLOBAL uint32_t*)spare : h.words;
  return h;
}

// The packed dwords a unit of four values needs, fetched but not yet consumed.
struct UnitRaw {
  u32x3_a4 d;
  uint32_t meta;  // bits 0..4: bit offset inside d.x; bits 8..9: kind; bits 16..: segment index
};
constexpr uint32_t UK_NONE = 0, UK_PACKED = 1, UK_RLE = 2, UK_SLOW = 3;

// For every unit (four consecutive values, unit k of lane t starts at value v0 + 4 * (k * 1024 + t))
// of the chunk's values [v0, v1): find its segment, classify it and ISSUE the 12-byte load of its
// packed dwords.  Nothing waits on the loads here (they complete behind the sweep of the page
// before), and every lane issues every load (units with nothing to read take dwords 0..2): a load
// whose destination is also written on another control-flow path would have to be waited for on
// the spot.  Units that straddle segments, lie in a segment cut short by the end of the stream,
// or hold ids wider than 16 bits are marked for the value-by-value path.
template <int UNITS>
__device__ __forceinline__ void locate_fetch(const PQD_LDS RleSeg* cs, const PageHead& h, uint32_t tid, uint32_t wave,
                                             UnitRaw (&raw)[UNITS] PROF_FN_PARAM) {
  const PQD_LDS uint8_t* table = (const PQD_LDS uint8_t*)(cs + SEG_TABLE_SLOT);
  const uint32_t t1 = h.v1;
  const uint32_t t1x4 = t1 << 2;
  // Every step below is done for all units before the next one, so that a step costs one LDS
  // round trip for the lane, not one per unit.
  uint32_t sg[UNITS], wi[UNITS];
  // the unit row of a wave is one block of the table; from its first segment walk forward
#pragma unroll
  for (int k = 0; k < UNITS; ++k) sg[k] = table[(uint32_t)k * (DS_THREADS * 4 / SEG_BLOCK) + wave];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    uint32_t nx[UNITS];
#pragma unroll
    for (int k = 0; k < UNITS; ++k) nx[k] = cs[sg[k] + 1].start4;
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
      const uint32_t g4 = 4 * (h.v0 + 4 * ((uint32_t)k * DS_THREADS + tid)) + 3;  // start4 <= g4  <=>  start <= g0
      sg[k] += nx[k] <= g4 ? 1u : 0u;                                 // valid lanes stop at the sentinel
    }
  }
  for (;;) {  // rare: more than two segment boundaries inside 256 values
    bool any = false;
    uint32_t nx[UNITS];
#pragma unroll
    for (int k = 0; k < UNITS; ++k) nx[k] = cs[sg[k] + 1].start4;
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
      const uint32_t g0 = h.v0 + 4 * ((uint32_t)k * DS_THREADS + tid);
      const bool adv = g0 < t1 && nx[k] <= g0 * 4 + 3;
      sg[k] += adv ? 1u : 0u;
      any |= adv;
    }
    if (!__builtin_amdgcn_ballot_w64(any)) break;
  }
  uint32_t sk[UNITS], srcs[UNITS], e4s[UNITS];
#pragma unroll
  for (int k = 0; k < UNITS; ++k) {
    sk[k] = cs[sg[k]].start4;
    srcs[k] = cs[sg[k]].src;
    e4s[k] = cs[sg[k] + 1].start4;
  }
#pragma unroll
  for (int k = 0; k < UNITS; ++k) {
    const uint32_t g0 = h.v0 + 4 * ((uint32_t)k * DS_THREADS + tid);
    const uint32_t lim4 = e4s[k] < t1x4 ? e4s[k] : t1x4;
    const bool is_packed = (sk[k] & SEG_PACKED_BIT) != 0;
    const bool whole = g0 * 4 + 16 <= lim4 && !(sk[k] & SEG_PARTIAL_BIT);
    // a segment holds at most PLAN_SEG_MAX_BYTES of payload: the bit offset inside it fits 32 bits
    const uint32_t byte0 = h.in_off + srcs[k];
    const uint32_t bitrel = (byte0 & 3u) * 8 + (g0 - (sk[k] >> 2)) * h.w;
    const uint32_t w_i = (byte0 >> 2) + (bitrel >> 5);
    const bool fast = whole && is_packed && !h.no_fast && w_i <= h.max_wi;
    const uint32_t kind = g0 >= t1 ? UK_NONE : (is_packed ? (fast ? UK_PACKED : UK_SLOW) : (whole ? UK_RLE : UK_SLOW));
    wi[k] = kind == UK_PACKED ? w_i : 0u;
    raw[k].meta = (bitrel & 31u) | (kind << 8) | (sg[k] << 16);
  }
  PROF(11)
#pragma unroll
  for (int k = 0; k < UNITS; ++k) raw[k].d = *(const PQD_GLOBAL u32x3_a4*)(h.load_words + wi[k]);
}

__device__ __forceinline__ uint32_t guarded_word(const PageHead& h, uint32_t i) {
  if (i < h.full_words) return h.words[i];
  if (i == h.full_words && h.tail_mask) return h.words[i] & h.tail_mask;
  return 0u;  // at or past the end of the stream
}

// Value j of a unit whose 64 aligned bits are {hi, lo}; w <= 16 and j are wave-uniform.
__device__ __forceinline__ uint32_t unit_value(uint32_t lo, uint32_t hi, uint32_t w, uint32_t j) {
  const uint32_t off = j * w;
  if (off + w <= 32) return __builtin_amdgcn_ubfe(lo, off, w);
  if (off >= 32) return __builtin_amdgcn_ubfe(hi, off - 32, w);
  return funnel_shr(lo, hi, off) & ((1u << w) - 1u);
}

// Turns the fetched dwords into ids (values at or beyond t1 become 0).  Units that straddle runs, lie
// in a run cut short by the end of the stream, or hold ids wider than 16 bits are not decoded by
// their lane: the wave collects them in its LDS list (`slow`: DS_SLOW_UNITS entries, then their
// ids) and decodes them one VALUE per lane, 16 units per step, so that pages with many short runs
// do not walk every lane through the value-by-value code for each of its units.
template <int UNITS>
__device__ __forceinline__ void extract(const PQD_LDS RleSeg* cs, const PageHead& h, uint32_t tid,
                                        const UnitRaw (&raw)[UNITS], uint32_t (&id)[UNITS * 4], PQD_LDS uint32_t* slow) {
  const uint32_t t1 = h.v1;
  const uint32_t w = h.w <= 16 ? h.w : 0u;  // wider ids never take the 12-byte path
  uint32_t slow_bits = 0;  // bit k: unit k of this lane is left to the list
#pragma unroll
  for (int k = 0; k < UNITS; ++k) {
    const uint32_t kind = (raw[k].meta >> 8) & 3u;
    const uint32_t b = raw[k].meta & 31u;
    const uint32_t lo = funnel_shr(raw[k].d.x, raw[k].d.y, b), hi = funnel_shr(raw[k].d.y, raw[k].d.z, b);
    const uint32_t a0 = unit_value(lo, hi, w, 0), a1 = unit_value(lo, hi, w, 1), a2 = unit_value(lo, hi, w, 2),
                   a3 = unit_value(lo, hi, w, 3);
    if (__builtin_amdgcn_ballot_w64(kind != UK_PACKED) == 0) {  // the usual case: the wave's whole row is packed
      id[4 * k] = a0;
      id[4 * k + 1] = a1;
      id[4 * k + 2] = a2;
      id[4 * k + 3] = a3;
    } else {
      const bool packed = kind == UK_PACKED;
      const uint32_t other = kind == UK_RLE ? cs[raw[k].meta >> 16].src : 0u;  // the repeated value of an RLE run
      id[4 * k] = packed ? a0 : other;
      id[4 * k + 1] = packed ? a1 : other;
      id[4 * k + 2] = packed ? a2 : other;
      id[4 * k + 3] = packed ? a3 : other;
      slow_bits |= kind == UK_SLOW ? 1u << k : 0u;
    }
  }
  if (__builtin_amdgcn_ballot_w64(slow_bits != 0) == 0) return;
  const uint32_t lane = tid & 63u, wave_tid0 = tid & ~63u;
  const uint64_t below = (1ull << lane) - 1ull;
  const uint32_t mask = mask32(h.w);
  PQD_LDS uint32_t* list = slow;
  PQD_LDS uint32_t* res = slow + DS_SLOW_UNITS;
  for (;;) {  // rounds of up to DS_SLOW_UNITS units (wave-uniform)
    uint32_t base = 0, mypos[UNITS];
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
      const bool pending = (slow_bits >> k) & 1u;
      const uint64_t m = __builtin_amdgcn_ballot_w64(pending);
      const uint32_t pos = base + (uint32_t)__builtin_popcountll(m & below);
      mypos[k] = pending && pos < (uint32_t)DS_SLOW_UNITS ? pos : ~0u;
      if (mypos[k] != ~0u) list[pos] = lane | ((uint32_t)k << 6) | ((raw[k].meta >> 16) << 16);
      base += (uint32_t)__builtin_popcountll(m);
    }
    const uint32_t n = base < (uint32_t)DS_SLOW_UNITS ? base : (uint32_t)DS_SLOW_UNITS;
    for (uint32_t b0 = 0; b0 < n; b0 += 16) {
      const uint32_t idx = b0 + (lane >> 2), j = lane & 3u;
      if (idx < n) {
        const uint32_t e = list[idx];
        const uint32_t g = h.v0 + 4 * (((e >> 6) & 7u) * DS_THREADS + wave_tid0 + (e & 63u)) + j;
        uint32_t t = e >> 16, v = 0;
        if (g < t1) {
          while (cs[t + 1].start4 <= g * 4 + 3) ++t;  // stops at the sentinel (start == v1 > g)
          const uint32_t tk = cs[t].start4, src = cs[t].src;
          v = src;
          if (tk & SEG_PACKED_BIT) {
            const uint32_t byte0 = h.in_off + src;
            const uint32_t bitrel = (byte0 & 3u) * 8 + (g - (tk >> 2)) * h.w;
            const uint32_t wi = (byte0 >> 2) + (bitrel >> 5);
            v = funnel_shr(guarded_word(h, wi), guarded_word(h, wi + 1), bitrel & 31u) & mask;
          }
        }
        res[idx * 4 + j] = v;
      }
    }
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
      if (mypos[k] != ~0u) {
        const u32x4 r = *(const PQD_LDS u32x4*)(res + mypos[k] * 4);
        id[4 * k] = r.x;
        id[4 * k + 1] = r.y;
        id[4 * k + 2] = r.z;
        id[4 * k + 3] = r.w;
        slow_bits &= ~(1u << k);
      }
    }
    if (base <= (uint32_t)DS_SLOW_UNITS) break;
  }
}

// One lookup in a quarter of a dictionary plane (8-byte dictionaries swept as dword planes, see
// pqd_rle_plan.hpp): the lanes whose id lies in the quarter being swept read 4 bytes at LDS address
// `addr` into `dst`, the others keep what they have.  The quarter of every id (bits 15..14) is kept
// as one byte, four values to a register (`tags`); a byte-selecting compare against the pass's
// quarter puts the matching lanes into EXEC, so a lookup is one vector instruction, one LDS read and
// one scalar move, with no branch and no wait.  `all` = the lanes active around the call.
template <int J>
__device__ __forceinline__ void plane_lookup(uint32_t& dst, uint32_t addr, uint32_t tags, uint32_t quarter, uint64_t all) {
#define PQD_PLANE_LOOKUP(SEL)                                                                              \
  asm volatile("v_cmpx_eq_u32_sdwa vcc, %2, %3 src0_sel:" SEL " src1_sel:DWORD\n\tds_read_b32 %0, %1\n\ts_mov_b64 exec, %4" \
               : "+v"(dst)                                                                                 \
               : "v"(addr), "v"(tags), "s"(quarter), "s"(all)                                              \
               : "vcc")
  if (J == 0) PQD_PLANE_LOOKUP("BYTE_0");
  if (J == 1) PQD_PLANE_LOOKUP("BYTE_1");
  if (J == 2) PQD_PLANE_LOOKUP("BYTE_2");
  if (J == 3) PQD_PLANE_LOOKUP("BYTE_3");
#undef PQD_PLANE_LOOKUP
}

// `between(k)` is called in front of the lookups of unit row k (a hook for work to interleave).
template <int R, typename Between>
__device__ __forceinline__ void plane_lookups(uint32_t (&dst)[R], const uint32_t (&addr)[R], const uint32_t (&tags)[R / 4],
                                              uint32_t quarter, uint64_t all, Between between) {
  const uint32_t q = uni(quarter);  // wave-uniform by construction; made so for the compiler
#pragma unroll
  for (int k = 0; k < R / 4; ++k) {
    between(k);
    plane_lookup<0>(dst[4 * k], addr[4 * k], tags[k], q, all);
    plane_lookup<1>(dst[4 * k + 1], addr[4 * k + 1], tags[k], q, all);
    plane_lookup<2>(dst[4 * k + 2], addr[4 * k + 2], tags[k], q, all);
    plane_lookup<3>(dst[4 * k + 3], addr[4 * k + 3], tags[k], q, all);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads above are invisible to the compiler
}

// Lookups of a chunk in an 8-byte dictionary (id[r] = the ids, below the dictionary's length).
template <int R, int UNITS>
__device__ __forceinline__ void sweep_8_byte_dictionary(uint32_t (&id)[R], uint32_t (&vlo)[R], uint32_t (&vhi)[R],
                                                        uint32_t dict_len, uint32_t slot1,
                                                        const uint8_t* plane_area, uint32_t lds0, uint32_t wave, uint32_t lane,
                                                        uint32_t& tag_s0, uint32_t& tag_s1, bool& descending PROF_FN_PARAM) {
  {
    // Plane sweep: pass s = plane * 4 + quarter brings 64 KiB (16384 entries x 4 bytes) of the low-
    // or high-dword plane into LDS buffer (quarter & 1), so an id's LDS address is (4 * id) & 0x1FFFF
    // in every pass.  Pass order alternates between chunks, as for slices: the two passes a chunk
    // ends with are the two the next one starts with.  A dictionary of one quarter per plane (at most
    // 16384 entries) keeps its low dwords in buffer 0 and its high dwords in buffer 1, both resident
    // from chunk to chunk; the addresses then carry the buffer bit while the high plane is read.
    const uint32_t entries = dict_len < 65536u ? dict_len : 65536u;
    const uint32_t nq = (entries + DP_QUARTER_ENTRIES - 1) / DP_QUARTER_ENTRIES;  // quarters per plane: 1..4
    const uint32_t NP = 2 * nq;
    const uint8_t* slot_base = plane_area + (size_t)(slot1 - 1u) * DP_SLOT_BYTES;
    uint32_t tags[R / 4];  // byte j of tags[k]: the quarter of value 4 * k + j
#pragma unroll
    for (int k = 0; k < R / 4; ++k) {
      tags[k] = (id[4 * k] >> 14) | ((id[4 * k + 1] >> 14) << 8) | ((id[4 * k + 2] >> 14) << 16) | ((id[4 * k + 3] >> 14) << 24);
#pragma unroll
      for (int j = 0; j < 4; ++j) id[4 * k + j] = (id[4 * k + j] << 2) & 0x1FFFFu;
    }
    const uint64_t all = __builtin_amdgcn_ballot_w64(true);
    auto pass_of = [&](uint32_t i) {  // sweep position -> plane * 4 + quarter
      const uint32_t idx = descending ? NP - 1 - i : i;
      const uint32_t pl = idx >= nq ? 1u : 0u;
      return pl * 4u + (idx - pl * nq);
    };
    auto load_quarter = [&](uint32_t ps) {
      const uint32_t qt = ps & 3u;
      const uint32_t n_here = entries - qt * DP_QUARTER_ENTRIES < DP_QUARTER_ENTRIES ? entries - qt * DP_QUARTER_ENTRIES : DP_QUARTER_ENTRIES;
      load_slice(slot_base + (size_t)(ps >> 2) * DP_PLANE_BYTES + (size_t)qt * DS_SLICE_BYTES, n_here * 4u,
                 lds0 + (nq == 1 ? ps >> 2 : qt & 1u) * DS_SLICE_BYTES, wave, lane);
    };
    bool prefetch_only = true;
    bool high_addresses = false;
    for (uint32_t i = 0; i < NP; ++i) {
      const uint32_t ps = pass_of(i);
      const uint32_t curb = nq == 1 ? ps >> 2 : ps & 1u;
      if ((nq == 1 && (ps >> 2) != 0) != high_addresses) {
        high_addresses = !high_addresses;
#pragma unroll
        for (int r = 0; r < R; ++r) id[r] ^= DS_SLICE_BYTES;
      }
      // tags of plane passes are kept apart from slice numbers by bit 8
      if ((curb ? tag_s1 : tag_s0) != (ps | 0x100u)) {
        wg_barrier();  // nobody still reads this buffer
        load_quarter(ps);
        if (curb)
          tag_s1 = ps | 0x100u;
        else
          tag_s0 = ps | 0x100u;
        prefetch_only = false;
      }
      PROF(5)
      if (prefetch_only)
        wg_barrier_keep<UNITS>();  // the prefetch loads stay in flight
      else
        wg_barrier();
      prefetch_only = false;
      PROF(3)
      // the next pass's quarter, if it goes to the other buffer, is requested in front of the lookups
      uint32_t nx = 0;
      bool bring_next = false;
      if (i + 1 < NP) {
        nx = pass_of(i + 1);
        if ((nq == 1 ? nx >> 2 : nx & 1u) != curb && (curb ? tag_s0 : tag_s1) != (nx | 0x100u)) {
          bring_next = true;
          if (curb)
            tag_s0 = nx | 0x100u;
          else
            tag_s1 = nx | 0x100u;
        }
      }
      // (issuing the transfer piece by piece between the lookups (`between`),
      // or behind them, were measured: 396 and 348 against 332-343 us per 100 M values: the waves queue
      // for the memory pipeline either way, and LDS reads run slower beside an LDS-DMA burst wherever it is put)
      if (bring_next) load_quarter(nx);
      auto between = [&](int) {};
      PROF(9)
      if (ps < 4u)
        plane_lookups(vlo, id, tags, ps & 3u, all, between);
      else
        plane_lookups(vhi, id, tags, ps & 3u, all, between);
      PROF(4)
    }
    descending = !descending;
  }
}

// ---------------------------------------------------------------------------------------------
// The gather for batches in which EVERY chunk is regular: nothing but whole bit-packed runs, one
// behind the other at a fixed distance (ids of a high-cardinality column as parquet-java writes them:
// 504-value runs behind one-byte headers; the plan kernel marks such chunks in their header and counts
// the others in counter 3 of the call's counter block).  The run of a unit then follows from its index,
// so the front end of the general kernel (chunk records in LDS, block table, segment search,
// classification, the list of straddling units, the chunk-end barrier) is not needed: a chunk is its
// 64-byte header and first segment, read by 18 lanes of every wave two chunks ahead.  The dictionaries
// are those of more than three quarters per plane (the plan kernel counts chunks of smaller ones with the
// irregular ones), which makes the sweep eight passes known at compile time: the quarter of an id is kept
// as two lane masks in scalar registers and a lookup costs no vector instruction; and because no chunk
// record travels by LDS-DMA, passes 0 and 1 (quarters already resident) need no wait for memory
// operations, so the stores of the chunk before drain behind them instead of in front of them.

struct RegHead {
  const PQD_GLOBAL uint32_t* words;  // the page's stream rounded down to a dword
  uint32_t in_off, max_wi;           // data & 3; largest dword index a 12-byte unit load may start at
  uint32_t v0, v1, w;
  uint32_t run_values, run_stride, start0, src0;
  float rcp_values;
};

// Dword `lane` of the chunk's header (lanes 0..15) and its first segment (lanes 16, 17).
__device__ __forceinline__ uint32_t load_chunk_head(const RleSeg* pool, uint32_t chunk, bool valid, uint32_t lane) {
  const PQD_GLOBAL uint32_t* base = (const PQD_GLOBAL uint32_t*)(pool + (size_t)chunk * SEG_CHUNK);
  const uint32_t idx = lane < 16u ? (uint32_t)SEG_HEAD_SLOT * 2u + lane : (lane < 18u ? lane - 16u : 0u);
  return valid ? base[idx] : 0u;  // all zero: an empty chunk
}
#define PQD_HEAD_DWORD(hv, i) ((uint32_t)__builtin_amdgcn_readlane((int)(hv), (i)))

__device__ __forceinline__ RegHead read_reg_head(uint32_t hv) {
  RegHead h;
  const uint32_t dlo = PQD_HEAD_DWORD(hv, 0), dhi = PQD_HEAD_DWORD(hv, 1);
  h.in_off = dlo & 3u;
  h.words = (const PQD_GLOBAL uint32_t*)((((uint64_t)dhi << 32) | dlo) - h.in_off);
  const uint32_t data_len = PQD_HEAD_DWORD(hv, 8);
  const uint32_t last_word = data_len ? (h.in_off + data_len - 1u) >> 2 : 0u;
  h.max_wi = last_word >= 2u ? last_word - 2u : 0u;  // (a regular chunk's page holds at least three dwords)
  h.run_values = PQD_HEAD_DWORD(hv, 14);
  // (a chunk that is not regular is one of the few the plan kernel listed for the left-over launch: it reads as empty)
  h.v0 = h.run_values ? PQD_HEAD_DWORD(hv, 4) : 0u;
  h.v1 = h.run_values ? PQD_HEAD_DWORD(hv, 5) : 0u;
  h.w = PQD_HEAD_DWORD(hv, 7);
  h.run_stride = PQD_HEAD_DWORD(hv, 15);
  h.start0 = PQD_HEAD_DWORD(hv, 16) >> 2;
  h.src0 = PQD_HEAD_DWORD(hv, 17);
  h.rcp_values = 1.0f / (float)(h.run_values ? h.run_values : 1u);
  return h;
}

// Issues the 12-byte loads of a regular chunk's units.  meta: bits 0..4 the bit offset inside the first
// dword loaded, bits 8..9 by how many dwords the load was moved down to stay inside the page.
// (`valid` false: nothing to fetch; the loads are issued all the same, from `spare`, so that their number is fixed.)
template <int UNITS>
__device__ __forceinline__ void fetch_regular(const RegHead& h, bool valid, const void* spare, uint32_t tid, u32x3_a4 (&raw)[UNITS],
                                              uint32_t (&meta)[UNITS]) {
  uint32_t wi[UNITS];
#pragma unroll
  for (int k = 0; k < UNITS; ++k) {
    const uint32_t g0 = h.v0 + 4 * ((uint32_t)k * DS_THREADS + tid);
    const uint32_t rel = (g0 < h.v1 ? g0 : h.v0) - h.start0;  // far below 2^24: exact as a float
    uint32_t q = (uint32_t)((float)rel * h.rcp_values);
    q -= q * h.run_values > rel ? 1u : 0u;
    q += (q + 1u) * h.run_values <= rel ? 1u : 0u;
    const uint32_t byte0 = h.in_off + h.src0 + q * h.run_stride;
    const uint32_t bitrel = (byte0 & 3u) * 8 + (rel - q * h.run_values) * h.w;
    const uint32_t w_i = (byte0 >> 2) + (bitrel >> 5);
    const uint32_t delta = w_i > h.max_wi ? w_i - h.max_wi : 0u;  // the page's last dwords: load what is there
    wi[k] = valid ? w_i - delta : 0u;
    meta[k] = (bitrel & 31u) | (delta << 8);
  }
  const PQD_GLOBAL uint32_t* words = valid ? h.words : (const PQD_GLOBAL uint32_t*)spare;
#pragma unroll
  for (int k = 0; k < UNITS; ++k) raw[k] = *(const PQD_GLOBAL u32x3_a4*)(words + wi[k]);
}

// One lookup in the quarter QT of a dictionary plane: the quarter of every id is kept as two lane masks
// (bit 15 and bit 14 of the id, computed once per chunk), one scalar instruction puts the quarter's lanes
// into EXEC, one 4-byte LDS read fills their result register: no vector arithmetic per lookup.  (All 64
// lanes of every wave are active around the call.)
template <int QT>
__device__ __forceinline__ void mask_lookup(uint32_t& dst, uint32_t addr, uint64_t hi, uint64_t lo) {
#define PQD_MASK_LOOKUP(SETEXEC) \
  asm volatile(SETEXEC "\n\tds_read_b32 %0, %1\n\ts_mov_b64 exec, -1" : "+v"(dst) : "v"(addr), "s"(hi), "s"(lo) : "scc")
  if (QT == 0) PQD_MASK_LOOKUP("s_nor_b64 exec, %2, %3");
  if (QT == 1) PQD_MASK_LOOKUP("s_andn2_b64 exec, %3, %2");
  if (QT == 2) PQD_MASK_LOOKUP("s_andn2_b64 exec, %2, %3");
  if (QT == 3) PQD_MASK_LOOKUP("s_and_b64 exec, %2, %3");
#undef PQD_MASK_LOOKUP
}

// Quarter `ps & 3` of plane `ps >> 2` of a four-quarter dictionary into LDS buffer (quarter & 1).
__device__ __forceinline__ void load_plane_quarter(const uint8_t* slot_base, uint32_t entries, uint32_t ps, uint32_t lds0,
                                                   uint32_t wave, uint32_t lane) {
  const uint32_t qt = ps & 3u;
  const uint32_t n_here = entries - qt * DP_QUARTER_ENTRIES < DP_QUARTER_ENTRIES ? entries - qt * DP_QUARTER_ENTRIES : DP_QUARTER_ENTRIES;
  load_slice(slot_base + (size_t)(ps >> 2) * DP_PLANE_BYTES + (size_t)qt * DS_SLICE_BYTES, n_here * 4u,
             lds0 + (qt & 1u) * DS_SLICE_BYTES, wave, lane);
}

// Pass Q (0..NP-1 in sweep order, NP = 4 for a 4-byte dictionary, which is its own single plane, 8 for the
// two planes of an 8-byte one; DESC: from the last quarter of the last plane down) of the sweep of a
// four-quarter dictionary.  Passes 0 and 1 find their quarters resident (the sweep before ended with
// them, in the other direction).  Barriers: none in front of pass 0; for LDS reads in front of pass 1
// (the quarter of pass 2 is about to overwrite the buffer pass 0 read); a full one from pass 2 on (the
// quarter requested one pass ago has landed, and everybody is done with the other buffer).
//   The next chunk's packed dwords are requested in pass NP - 2, right behind that pass's quarter
// request (`prefetch` issues exactly KEEP loads), and the barrier of the pass after it leaves those KEEP
// loads in flight (the counters are in issue order): requested in front of the sweep they would be
// waited for at pass 2's barrier, a microsecond later, which is less than a load from HBM takes while
// the chip is writing.
template <int R, int NP, int Q, bool DESC, int KEEP, typename Prefetch>
__device__ __forceinline__ void regular_pass(const uint8_t* slot_base, uint32_t entries, uint32_t lds0, uint32_t wave, uint32_t lane,
                                             uint32_t (&vlo)[R], uint32_t (&vhi)[NP == 8 ? R : 1], const uint32_t (&addr)[R],
                                             const uint64_t (&mhi)[R], const uint64_t (&mlo)[R], Prefetch prefetch) {
  constexpr int S = DESC ? NP - 1 - Q : Q;  // plane S >> 2, quarter S & 3
  constexpr int PL = S >> 2, QT = S & 3;
  if (Q == 1) wg_barrier_lds();
  if (Q >= 2 && Q != NP - 1) wg_barrier();
  if (Q >= 2 && Q == NP - 1) wg_barrier_keep<KEEP>();
  if (Q >= 1 && Q + 1 < NP) load_plane_quarter(slot_base, entries, (uint32_t)(DESC ? S - 1 : S + 1), lds0, wave, lane);
  if (Q == NP - 2) prefetch();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (PL == 0)
      mask_lookup<QT>(vlo[r], addr[r], mhi[r], mlo[r]);
    else
      mask_lookup<QT>(vhi[NP == 8 ? r : 0], addr[r], mhi[r], mlo[r]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads above are invisible to the compiler
}

template <int R, int NP, bool DESC, int KEEP, typename Prefetch>
__device__ __forceinline__ void regular_sweep(const uint8_t* slot_base, uint32_t entries, uint32_t lds0, uint32_t wave, uint32_t lane,
                                              uint32_t (&vlo)[R], uint32_t (&vhi)[NP == 8 ? R : 1], const uint32_t (&addr)[R],
                                              const uint64_t (&mhi)[R], const uint64_t (&mlo)[R], Prefetch prefetch) {
  regular_pass<R, NP, 0, DESC, KEEP>(slot_base, entries, lds0, wave, lane, vlo, vhi, addr, mhi, mlo, prefetch);
  regular_pass<R, NP, 1, DESC, KEEP>(slot_base, entries, lds0, wave, lane, vlo, vhi, addr, mhi, mlo, prefetch);
  regular_pass<R, NP, 2, DESC, KEEP>(slot_base, entries, lds0, wave, lane, vlo, vhi, addr, mhi, mlo, prefetch);
  regular_pass<R, NP, 3, DESC, KEEP>(slot_base, entries, lds0, wave, lane, vlo, vhi, addr, mhi, mlo, prefetch);
  if constexpr (NP == 8) {
    regular_pass<R, NP, 4, DESC, KEEP>(slot_base, entries, lds0, wave, lane, vlo, vhi, addr, mhi, mlo, prefetch);
    regular_pass<R, NP, 5, DESC, KEEP>(slot_base, entries, lds0, wave, lane, vlo, vhi, addr, mhi, mlo, prefetch);
    regular_pass<R, NP, 6, DESC, KEEP>(slot_base, entries, lds0, wave, lane, vlo, vhi, addr, mhi, mlo, prefetch);
    regular_pass<R, NP, 7, DESC, KEEP>(slot_base, entries, lds0, wave, lane, vlo, vhi, addr, mhi, mlo, prefetch);
  }
}

// The whole batch, for one workgroup of k_dict_pages_sliced (which calls it when counter 3 of the call's
// counter block is zero: no chunk of the batch is irregular).  `lds0`: LDS byte address of the dynamic block.
// RESIDENT: every dictionary of the batch fits the two LDS buffers and stays there (no sweep, no masks, no barrier
// between chunks); otherwise every dictionary is swept in four quarters per plane.  The plan kernel marks resident
// chunks for 4-byte dictionaries only: with 8-byte values the waves, which nothing keeps in step here, write their
// output 25 % slower than the general path does (measured: 385 us against 303 per 100 M values, 337 with a barrier per chunk).  (One body for both kinds, chosen
// per chunk, costs the swept kind registers it does not have; the plan kernel tells the kinds apart per batch.)
template <int ELEM, bool RESIDENT>
__device__ __forceinline__ void gather_regular_batch(pqd_page_result* __restrict__ results, const RleSeg* __restrict__ pool,
                                                     uint32_t pool_chunks, const uint32_t* __restrict__ cursor, uint32_t n_pages,
                                                     const uint8_t* __restrict__ plane_area, uint32_t lds0, uint32_t stagger) {
  constexpr int R = DS_R;
  constexpr int UNITS = R / 4;
  constexpr int NP = ELEM == 8 ? 8 : 4;  // passes per sweep
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = uni(tid >> 6), lane = tid & 63u;

  uint64_t tag_dict = 0;  // the dictionary whose quarters the LDS buffers hold
  uint32_t tag_len = 0;
  bool descending = false;

  // workgroup b takes chunks b, b + grid, b + 2 * grid, ...: at any moment the workgroups of the chip
  // read and write one window of the batch that moves along (PQD_REGULAR_SHARES: contiguous shares instead)
  uint32_t n_chunks = n_pages + uni(cursor[0]);
  if (n_chunks > pool_chunks) n_chunks = pool_chunks;
  // a batch without a single chunk of this kind (every page went to the left-over launch, decoded whole or gathered from
  // memory): nothing to visit.  (As a select on the count: a branch here measured 1.3 % on every other batch, through
  // nothing but a different register allocation; for 4-byte values even the select cost 1.3 %, so they visit the empty
  // chunks: 36 us per 5000 pages.)
  if (ELEM == 8 && uni(cursor[RESIDENT ? PLAN_CTR_RESIDENT : PLAN_CTR_SWEPT]) == 0u) n_chunks = 0u;
  const uint32_t share = (n_chunks + gridDim.x - 1) / gridDim.x;
#ifdef PQD_REGULAR_SHARES
  uint32_t chunk = blockIdx.x * share;
  if (chunk >= n_chunks) return;
  n_chunks = chunk + share < n_chunks ? chunk + share : n_chunks;  // from here on: the end of this workgroup's share
  const uint32_t step = 1;
#else
  uint32_t chunk = blockIdx.x;
  if (chunk >= n_chunks) return;
  const uint32_t step = gridDim.x;
#endif

  uint32_t hv = load_chunk_head(pool, chunk, true, lane);                             // this chunk
  uint32_t hv1 = load_chunk_head(pool, chunk + step, chunk + step < n_chunks, lane);  // the next one
  u32x3_a4 raw[UNITS];
  uint32_t meta[UNITS];
  {
    const RegHead h0 = read_reg_head(hv);
    // staggered start of long batches over large dictionaries (see the general kernel)
    if (!RESIDENT && share >= 16u) {
      const uint32_t phase = (blockIdx.x >> 3) & 31u;
      for (uint32_t i = 0; i < phase * stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }
    fetch_regular<UNITS>(h0, h0.v1 > h0.v0, pool, tid, raw, meta);
  }

  for (;;) {
    const RegHead h = read_reg_head(hv);
    const uint32_t next_chunk = chunk + step;
    const bool has_next = next_chunk < n_chunks;
    // the header of the chunk after the next one is requested now; the next chunk's is here already
    const uint32_t hv2 = load_chunk_head(pool, next_chunk + step, has_next && next_chunk + step < n_chunks, lane);
    if (h.v1 > h.v0) {  // (uniform over the workgroup; an empty chunk belongs to a page the plan kernel finished or left over)
      const uint32_t v0 = h.v0, n_out = h.v1;
      const uint32_t dict_len = PQD_HEAD_DWORD(hv, 11);
      const uint64_t dict_addr = ((uint64_t)PQD_HEAD_DWORD(hv, 13) << 32) | PQD_HEAD_DWORD(hv, 12);
      const uint64_t out_addr = ((uint64_t)PQD_HEAD_DWORD(hv, 3) << 32) | PQD_HEAD_DWORD(hv, 2);
      const uint32_t page = PQD_HEAD_DWORD(hv, 10);
      const uint32_t slot1 = PQD_HEAD_DWORD(hv, 6) >> 16;
      const uint32_t entries = dict_len < 65536u ? dict_len : 65536u;  // resident, or more than three quarters (the plan kernel saw to it)
      // the planes of an 8-byte dictionary are its plane copy; a 4-byte dictionary is its own single plane
      const uint8_t* slot_base = ELEM == 8 ? plane_area + (size_t)(slot1 - 1u) * DP_SLOT_BYTES : reinterpret_cast<const uint8_t*>(dict_addr);
      // A dictionary that fits the two LDS buffers stays there: 4-byte entries as they are (up to 32768 of them),
      // 8-byte entries as their two planes (up to 16384 each).  Such chunks need no sweep, no masks and no barrier.
      constexpr bool resident = RESIDENT;
      if (dict_addr != tag_dict || dict_len != tag_len) {  // first chunk, another dictionary
        tag_dict = dict_addr;
        tag_len = dict_len;
        wg_barrier();  // nobody still reads the buffers
        if (resident) {
          descending = false;
          if (ELEM == 8) {
            load_slice(slot_base, entries * 4u, lds0, wave, lane);
            load_slice(slot_base + DP_PLANE_BYTES, entries * 4u, lds0 + DS_SLICE_BYTES, wave, lane);
          } else {
            load_plane_quarter(slot_base, entries, 0u, lds0, wave, lane);
            if (entries > DP_QUARTER_ENTRIES) load_plane_quarter(slot_base, entries, 1u, lds0, wave, lane);
          }
        } else {  // the two quarters the sweep starts with
          load_plane_quarter(slot_base, entries, descending ? (uint32_t)NP - 1u : 0u, lds0, wave, lane);
          load_plane_quarter(slot_base, entries, descending ? (uint32_t)NP - 2u : 1u, lds0, wave, lane);
        }
        wg_barrier();
      }

      // ---- 1. ids into registers (their packed dwords were requested one chunk ago)
      uint32_t id[R];
      const uint32_t w = h.w;
#pragma unroll
      for (int k = 0; k < UNITS; ++k) {
        uint32_t d0 = raw[k].x, d1 = raw[k].y, d2 = raw[k].z;
        const uint32_t dl = meta[k] >> 8;
        if (__builtin_amdgcn_ballot_w64(dl != 0)) {  // rare: a unit in the page's last dwords
          d0 = dl == 1 ? d1 : (dl == 2 ? d2 : d0);
          d1 = dl == 1 ? d2 : (dl == 2 ? 0u : d1);
          d2 = dl == 0 ? d2 : 0u;
        }
        const uint32_t b = meta[k] & 31u;
        const uint32_t lo = funnel_shr(d0, d1, b), hi = funnel_shr(d1, d2, b);
        id[4 * k] = unit_value(lo, hi, w, 0);
        id[4 * k + 1] = unit_value(lo, hi, w, 1);
        id[4 * k + 2] = unit_value(lo, hi, w, 2);
        id[4 * k + 3] = unit_value(lo, hi, w, 3);
      }

      // ---- 2. ids beyond the dictionary (see the general kernel)
      const uint32_t id_bits = (w + 7u) & ~7u;
      if (id_bits >= 32 || (1u << id_bits) > dict_len) {
        uint32_t bad = ~0u;
#pragma unroll
        for (int k = UNITS - 1; k >= 0; --k) {
          const uint32_t g0 = v0 + 4 * ((uint32_t)k * DS_THREADS + tid);
#pragma unroll
          for (int j = 3; j >= 0; --j) {
            const bool over = id[4 * k + j] >= dict_len;
            bad = (over && g0 + j < n_out) ? g0 + j : bad;
            id[4 * k + j] = over ? 0u : id[4 * k + j];
          }
        }
        if (bad != ~0u) {  // the plan kernel wrote the page's result; an id beyond the dictionary comes before whatever ended the header walk
          atomicMin(&results[page].values_written, bad);
          results[page].status = PQD_DICT_OOB;
        }
      }

      // ---- 3. (the next chunk's packed dwords are requested inside the sweep)
      const RegHead nh = read_reg_head(hv1);
      auto prefetch = [&]() { fetch_regular<UNITS>(nh, has_next && nh.v1 > nh.v0, pool, tid, raw, meta); };

      // ---- 4. gather
      uint32_t vlo[R], vhi[ELEM == 8 ? R : 1];
      if (resident) {
        prefetch();
        const PQD_LDS uint8_t* lds_bytes = (const PQD_LDS uint8_t*)(uintptr_t)lds0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          vlo[r] = *(const PQD_LDS uint32_t*)(lds_bytes + id[r] * 4u);
          if (ELEM == 8) vhi[ELEM == 8 ? r : 0] = *(const PQD_LDS uint32_t*)(lds_bytes + DS_SLICE_BYTES + id[r] * 4u);
        }
      } else {
        // an id's LDS address is (4 * id) & 0x1FFFF in every pass, its quarter two lane masks
        uint64_t mhi[R], mlo[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          vlo[r] = 0;
          vhi[ELEM == 8 ? r : 0] = 0;
          mhi[r] = __builtin_amdgcn_ballot_w64(id[r] >= 0x8000u);
          id[r] = (id[r] << 2) & 0x1FFFFu;
          mlo[r] = __builtin_amdgcn_ballot_w64(id[r] >= 0x10000u);
        }
        if (!descending)
          regular_sweep<R, NP, false, UNITS>(slot_base, entries, lds0, wave, lane, vlo, vhi, id, mhi, mlo, prefetch);
        else
          regular_sweep<R, NP, true, UNITS>(slot_base, entries, lds0, wave, lane, vlo, vhi, id, mhi, mlo, prefetch);
        descending = !descending;
      }

      // ---- 5. store, 16 bytes per lane per instruction
#pragma unroll
      for (int k = 0; k < UNITS; ++k) {
        const uint32_t g0 = v0 + 4 * ((uint32_t)k * DS_THREADS + tid);
        if (g0 >= n_out) continue;
        if constexpr (ELEM == 4) {
          PQD_GLOBAL uint32_t* o = (PQD_GLOBAL uint32_t*)out_addr + g0;
          if (g0 + 4 <= n_out) {
            u32x4_a4 r4 = {vlo[4 * k], vlo[4 * k + 1], vlo[4 * k + 2], vlo[4 * k + 3]};
            PQD_STREAM_STORE((PQD_GLOBAL u32x4_a4*)(o), r4);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (g0 + j < n_out) o[j] = vlo[4 * k + j];
          }
        } else {
          PQD_GLOBAL u32x2_a4* o = (PQD_GLOBAL u32x2_a4*)out_addr + g0;
          if (g0 + 4 <= n_out) {
            u32x4_a4 r0 = {vlo[4 * k], vhi[ELEM == 8 ? 4 * k : 0], vlo[4 * k + 1], vhi[ELEM == 8 ? 4 * k + 1 : 0]};
            u32x4_a4 r1 = {vlo[4 * k + 2], vhi[ELEM == 8 ? 4 * k + 2 : 0], vlo[4 * k + 3], vhi[ELEM == 8 ? 4 * k + 3 : 0]};
            PQD_STREAM_STORE((PQD_GLOBAL u32x4_a4*)(o), r0);
            PQD_STREAM_STORE((PQD_GLOBAL u32x4_a4*)(o + 2), r1);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (g0 + j < n_out) {
                u32x2_a4 x = {vlo[4 * k + j], vhi[ELEM == 8 ? 4 * k + j : 0]};
                o[j] = x;
              }
          }
        }
      }
    } else if (has_next) {
      const RegHead nh = read_reg_head(hv1);
      fetch_regular<UNITS>(nh, nh.v1 > nh.v0, pool, tid, raw, meta);
    }
    if (!has_next) break;
    hv = hv1;
    hv1 = hv2;
    chunk = next_chunk;
  }
}
#undef PQD_HEAD_DWORD

template <int ELEM>
__global__ __launch_bounds__(DS_THREADS) void k_dict_pages_sliced(pqd_page_result* __restrict__ results,
                                                                  const RleSeg* __restrict__ pool, uint32_t pool_chunks,
                                                                  const uint32_t* __restrict__ cursor, uint32_t n_pages,
                                                                  const uint8_t* __restrict__ plane_area,
                                                                  uint32_t regular_allowed PROF_PARAM) {
  extern __shared__ u32x4 lds4[];
  uint8_t* lds = reinterpret_cast<uint8_t*>(lds4);
  DsShared* sh = reinterpret_cast<DsShared*>(lds + DS_OFF_STATE);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(PQD_LDS uint8_t*)lds;  // LDS byte address of the dynamic block
  // a batch made of regular chunks only (the plan kernel counted no other) takes the path written for it
  if ((regular_allowed & 1u) && __builtin_amdgcn_readfirstlane(cursor[PLAN_CTR_IRREGULAR]) == 0u &&
      (__builtin_amdgcn_readfirstlane(cursor[PLAN_CTR_RESIDENT]) == 0u || __builtin_amdgcn_readfirstlane(cursor[PLAN_CTR_SWEPT]) == 0u)) {
    // (counters 7 / 8: some regular chunk has a swept / a resident dictionary; a batch with both kinds is flagged irregular)
    if (__builtin_amdgcn_readfirstlane(cursor[PLAN_CTR_RESIDENT]) != 0u)
      gather_regular_batch<ELEM, true>(results, pool, pool_chunks, cursor, n_pages, plane_area, lds0, regular_allowed >> 8);
    else
      gather_regular_batch<ELEM, false>(results, pool, pool_chunks, cursor, n_pages, plane_area, lds0, regular_allowed >> 8);
    return;
  }
  if (regular_allowed & 2u) return;  // PQD_DICT_REGULAR=only (tests): nothing but the regular path may write

  constexpr int R = DS_R;
  constexpr int UNITS = R / 4;
  constexpr uint32_t S = DS_SLICE_BYTES / ELEM;  // dictionary entries per slice
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = uni(tid >> 6), lane = tid & 63u;

  // which slice each LDS buffer holds (uniform over the workgroup)
  uint64_t tag_dict = 0;
  uint32_t tag_len = 0;
  uint32_t tag_s0 = ~0u, tag_s1 = ~0u;
  uint32_t last_buf = 1;   // buffer of the latest pass
  uint32_t res_win = ~0u;  // window mode: which window (0 / 1) the two buffers hold; ~0: none
  bool descending = false;  // slice order of the current page

  // chunks to take: one per page (empty ones included) and the extra ones the plan kernel took from
  // the shared tail (it may have asked for more than the pool holds)
  uint32_t n_chunks = n_pages + uni(*cursor);
  if (n_chunks > pool_chunks) n_chunks = pool_chunks;
  // every workgroup takes one contiguous share of the chunks, so that at any moment the workgroups
  // are spread over the whole batch
  const uint32_t share = (n_chunks + gridDim.x - 1) / gridDim.x;
  uint32_t chunk = blockIdx.x * share;
  if (chunk >= n_chunks) return;
  n_chunks = chunk + share < n_chunks ? chunk + share : n_chunks;  // from here on: the end of this workgroup's share
  const uint32_t stride = 1;
  // Chunk buffers form a ring of three: the current chunk, the next one (its packed dwords are
  // fetched while the current chunk is gathered) and the one after (it is on its way).
  uint32_t pb = 0;
  fetch_chunk_async(pool, chunk, lds0 + DS_OFF_SEGS, wave, lane);
  if (chunk + stride < n_chunks) fetch_chunk_async(pool, chunk + stride, lds0 + DS_OFF_SEGS + DS_SEG_BYTES, wave, lane);
  if (tid < 3) sh->oob_at[tid] = ~0u;
  wg_barrier();

  // Long batches of 8-byte dictionaries that are swept through LDS: the workgroups start a few
  // microseconds apart (32 phases of 0.4 us).  Left alone they run in step, chunk after chunk, so the whole
  // chip asks L2 for slices at one moment and writes its output at the next; out of step the two
  // overlap.  Measured on 100 M values, 64 Ki entries: 359 -> 334 us including the delay (eight phases of 1.7 us: 340); 4-byte
  // dictionaries gain nothing and short batches would only pay the delay, so they start at once.
  if (ELEM == 8 && share >= 16u) {
    const PQD_LDS RleChunkHead* hd0 = (const PQD_LDS RleChunkHead*)((const PQD_LDS RleSeg*)(lds + DS_OFF_SEGS) + SEG_HEAD_SLOT);
    if (uni(hd0->dict_len) > S) {
      const uint32_t phase = (blockIdx.x >> 3) & 31u;  // blockIdx & 7 is the XCD: its 32 workgroups get a phase each
      for (uint32_t i = 0; i < phase; ++i) __builtin_amdgcn_s_sleep(16);
    }
  }
  UnitRaw raw[UNITS];  // packed dwords of the chunk about to be decoded
  PROF_INIT
  {
    const PageHead h0 = read_head((const PQD_LDS RleSeg*)(lds + DS_OFF_SEGS), true, pool);
    locate_fetch<UNITS>((const PQD_LDS RleSeg*)(lds + DS_OFF_SEGS), h0, tid, wave, raw PROF_FN_ARG);
  }

  for (;;) {
    const uint32_t nb = pb == 2 ? 0u : pb + 1, nnb = nb == 2 ? 0u : nb + 1;
    const PQD_LDS RleSeg* cs = (const PQD_LDS RleSeg*)(lds + DS_OFF_SEGS + pb * DS_SEG_BYTES);
    const PQD_LDS RleChunkHead* hd = (const PQD_LDS RleChunkHead*)(cs + SEG_HEAD_SLOT);
    const PageHead h = read_head(cs, true, pool);
    const uint32_t v0 = h.v0, n_out = h.v1;  // output values [v0, n_out) of the page; at most DS_TILE_VALUES of them
    const uint32_t dict_len = uni(hd->dict_len);
    const uint64_t dict_addr = ((uint64_t)uni((uint32_t)(hd->dict >> 32)) << 32) | uni((uint32_t)hd->dict);
    const uint64_t out_addr = ((uint64_t)uni((uint32_t)(hd->out >> 32)) << 32) | uni((uint32_t)hd->out);
    const uint32_t page = uni(hd->page);
    const uint32_t next_chunk = chunk + stride;
    const bool has_next = next_chunk < n_chunks;
    const uint32_t P = (dict_len + S - 1) / S;
    const uint8_t* dict_bytes = reinterpret_cast<const uint8_t*>(dict_addr);
    // 8-byte dictionaries are looked up in dword planes: the slot of this one's plane copy
    const uint32_t slot1 = ELEM == 8 ? uni(hd->nseg) >> 16 : 0u;
    if (dict_addr != tag_dict || dict_len != tag_len) {  // another dictionary: nothing resident is of use
      tag_dict = dict_addr;
      tag_len = dict_len;
      tag_s0 = tag_s1 = ~0u;
      res_win = ~0u;
    }
    PROF(0)

    // ---- 1. ids into registers (their packed dwords were requested one page ago)
    uint32_t id[R];
    extract<UNITS>(cs, h, tid, raw, id, (PQD_LDS uint32_t*)(lds + DS_OFF_SLOW + wave * DS_SLOW_BYTES));
    PROF(1)

    // ---- 2. ids beyond the dictionary: note the first one, and look up entry 0 instead.
    // A page whose ids cannot express such an id skips the check.  What counts is the width in whole
    // bytes: the value of an RLE run is ceil(w/8) bytes and is not masked to w bits
    // (BytesUtils.readIntLittleEndianPaddedOnBitWidth), so a 3-bit page may well hold the id 200.
    const uint32_t id_bits = (h.w + 7u) & ~7u;
    if (id_bits >= 32 || (1u << id_bits) > dict_len) {
      uint32_t bad = ~0u;
#pragma unroll
      for (int k = UNITS - 1; k >= 0; --k) {
        const uint32_t g0 = v0 + 4 * ((uint32_t)k * DS_THREADS + tid);
#pragma unroll
        for (int j = 3; j >= 0; --j) {
          const bool over = id[4 * k + j] >= dict_len;
          bad = (over && g0 + j < n_out) ? g0 + j : bad;
          id[4 * k + j] = over ? 0u : id[4 * k + j];
        }
      }
      if (bad != ~0u) atomicMin(&sh->oob_at[pb], bad);
    }
    PROF(2)

    // ---- 3. gather.  Byte offsets into the dictionary, so that a pass costs a subtraction and a
    // compare per id; a pass looks up the ids that fall into the part of the dictionary in LDS.
    if constexpr (ELEM == 4) {
#pragma unroll
      for (int r = 0; r < R; ++r) id[r] *= ELEM;
    }
    uint32_t vlo[R], vhi[ELEM == 8 ? R : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      vlo[r] = 0;
      vhi[ELEM == 8 ? r : 0] = 0;
    }
    // ids whose dictionary byte offset lies in [base, base + limit) are looked up at lds[offset - base + at]
    auto lookups = [&](uint32_t base, uint32_t limit, uint32_t at) {
      const PQD_LDS uint8_t* buf = (const PQD_LDS uint8_t*)lds + at;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t a = id[r] - base;
        if (a < limit) {
          if (ELEM == 4) {
            vlo[r] = *(const PQD_LDS uint32_t*)(buf + a);
          } else {
            const u32x2 t = *(const PQD_LDS u32x2*)(buf + a);
            vlo[r] = t.x;
            vhi[ELEM == 8 ? r : 0] = t.y;
          }
        }
      }
    };
    // Dictionaries of up to four slices are handled in WINDOWS of two slices (the two LDS buffers
    // side by side): half as many passes over the ids.  With two windows, a page starts with the
    // window the page before ended with, so a page switches windows once; the switch (128 KiB of
    // LDS-DMA) runs behind the prefetch work for the next page.
    // (8-byte values keep to single slices: the extra code costs them registers they do not have)
    const bool wide = ELEM == 4 && P <= 4;
    const bool two_windows = P > 2 && wide;
    const uint32_t dict_bytes_total = dict_len * ELEM;
    auto load_window = [&](uint32_t win) {  // slices 2*win and 2*win + 1 into buffers 0 and 1
#pragma unroll
      for (uint32_t half = 0; half < 2; ++half) {
        const uint32_t sl = 2 * win + half;
        if (sl < P) {
          const uint32_t n_here = dict_len - sl * S < S ? dict_len - sl * S : S;
          load_slice(dict_bytes + (size_t)sl * DS_SLICE_BYTES, n_here * ELEM, lds0 + half * DS_SLICE_BYTES, wave, lane);
        }
      }
    };
    if (wide && P > 0 && res_win > 1u) {  // nothing of this dictionary is resident (first page, other dictionary)
      wg_barrier();  // nobody still reads the buffers
      load_window(0);
      res_win = 0;
      tag_s0 = tag_s1 = ~0u;  // the per-slice bookkeeping of the narrow mode no longer holds
      wg_barrier();
    }
    if (two_windows) {
      const uint32_t wbase = res_win * 2 * DS_SLICE_BYTES;
      lookups(wbase, 2 * DS_SLICE_BYTES, 0u);
      PROF(4)
      wg_barrier_lds();  // everybody is done with the resident window
      res_win ^= 1u;
      load_window(res_win);
      PROF(9)
    }

    // ---- 4. prefetch: the chunk after the next, the packed dwords of the next chunk.
    // The loads are issued unconditionally (from a harmless address when there is nothing to
    // fetch) so that exactly UNITS of them are this wave's most recent memory operations when it
    // reaches the next barrier.
    if (next_chunk + stride < n_chunks && next_chunk + stride >= next_chunk)
      fetch_chunk_async(pool, next_chunk + stride, lds0 + DS_OFF_SEGS + nnb * DS_SEG_BYTES, wave, lane);
    {
      // without a next chunk the buffer holds stale bytes: read_head then describes an empty range
      const PQD_LDS RleSeg* ncs = (const PQD_LDS RleSeg*)(lds + DS_OFF_SEGS + nb * DS_SEG_BYTES);
      const PageHead nh = read_head(ncs, has_next, pool);
      locate_fetch<UNITS>(ncs, nh, tid, wave, raw PROF_FN_ARG);
    }
    PROF(8)

    if (two_windows) {
      wg_barrier_keep<UNITS>();  // the other window has landed; the prefetch loads stay in flight
      PROF(3)
      lookups(res_win * 2 * DS_SLICE_BYTES, 2 * DS_SLICE_BYTES, 0u);
      PROF(4)
    } else if (wide) {
      if (P > 0) lookups(0u, dict_bytes_total, 0u);  // the whole dictionary is in LDS
      PROF(4)
    } else if constexpr (ELEM == 8) {
      res_win = ~0u;
      if (P > 0 && slot1 != 0)
        sweep_8_byte_dictionary<R, UNITS>(id, vlo, vhi, dict_len, slot1, plane_area, lds0, wave, lane,
                                          tag_s0, tag_s1, descending PROF_FN_ARG);
    } else {
      // More than four slices: one slice per pass, double-buffered.  Pass order alternates between
      // pages (0,1,..,P-1 then P-1,..,1,0): the two slices a page ends with are the two the next
      // page starts with, so a page moves only P - 2 slices into LDS.
      res_win = ~0u;
      bool prefetch_only = true;  // this wave's memory operations since the last barrier: the prefetch loads alone
      for (uint32_t i = 0; i < P; ++i) {
        const uint32_t s = descending ? P - 1 - i : i;
        uint32_t curb = tag_s0 == s ? 0u : (tag_s1 == s ? 1u : last_buf ^ 1u);
        // the slice this pass needs, if it is not resident
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
        last_buf = curb;
        PROF(5)
        // this pass's slice has landed; everybody is done with the other buffer
        if (prefetch_only)
          wg_barrier_keep<UNITS>();  // the prefetch loads stay in flight
        else
          wg_barrier();
        prefetch_only = false;
        PROF(3)
        // bring in the slice of the next pass
        if (i + 1 < P) {
          const uint32_t nx = descending ? s - 1 : s + 1;
          if (tag_s0 != nx && tag_s1 != nx) {
            const uint32_t n_here = dict_len - nx * S < S ? dict_len - nx * S : S;
            load_slice(dict_bytes + (size_t)nx * DS_SLICE_BYTES, n_here * ELEM, lds0 + (curb ^ 1u) * DS_SLICE_BYTES, wave, lane);
            if (curb)
              tag_s0 = nx;
            else
              tag_s1 = nx;
          }
        }
        PROF(9)
        lookups(s * DS_SLICE_BYTES, DS_SLICE_BYTES, curb * DS_SLICE_BYTES);
        PROF(4)
      }
      descending = !descending;
    }

    // ---- 5. store, 16 bytes per lane per instruction
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
      const uint32_t g0 = v0 + 4 * ((uint32_t)k * DS_THREADS + tid);
      if (g0 >= n_out) continue;
      if (ELEM == 4) {
        PQD_GLOBAL uint32_t* o = (PQD_GLOBAL uint32_t*)out_addr + g0;
        if (g0 + 4 <= n_out) {
          u32x4_a4 r4 = {vlo[4 * k], vlo[4 * k + 1], vlo[4 * k + 2], vlo[4 * k + 3]};
          PQD_STREAM_STORE((PQD_GLOBAL u32x4_a4*)(o), r4);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (g0 + j < n_out) o[j] = vlo[4 * k + j];
        }
      } else {
        PQD_GLOBAL u32x2_a4* o = (PQD_GLOBAL u32x2_a4*)out_addr + g0;
        if (g0 + 4 <= n_out) {
          u32x4_a4 r0 = {vlo[4 * k], vhi[ELEM == 8 ? 4 * k : 0], vlo[4 * k + 1], vhi[ELEM == 8 ? 4 * k + 1 : 0]};
          u32x4_a4 r1 = {vlo[4 * k + 2], vhi[ELEM == 8 ? 4 * k + 2 : 0], vlo[4 * k + 3], vhi[ELEM == 8 ? 4 * k + 3 : 0]};
          PQD_STREAM_STORE((PQD_GLOBAL u32x4_a4*)(o), r0);
          PQD_STREAM_STORE((PQD_GLOBAL u32x4_a4*)(o + 2), r1);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (g0 + j < n_out) {
              u32x2_a4 x = {vlo[4 * k + j], vhi[ELEM == 8 ? 4 * k + j : 0]};
              o[j] = x;
            }
        }
      }
    }
    PROF(6)

    // ---- 6. chunk end.  All atomicMin on this chunk's oob slot are in after the barrier; the slot
    // is next used three chunks on.  A chunk without a sweep has had no barrier that made sure the
    // chunk DMA issued above has landed: it takes the full one.
    if (P == 0 || (ELEM == 8 && slot1 == 0))
      wg_barrier();
    else if (wide && !two_windows)
      wg_barrier_keep<UNITS>();  // the only barrier of such a chunk: waits for everything but its last stores
    else
      wg_barrier_lds();
    if (tid == 0) {
      const uint32_t oob = sh->oob_at[pb];
      if (oob != ~0u) {
        // the plan kernel wrote the page's result; an id beyond the dictionary comes before whatever
        // ended the header walk.  Chunks of one page may report from different workgroups.
        sh->oob_at[pb] = ~0u;
        atomicMin(&results[page].values_written, oob);
        results[page].status = PQD_DICT_OOB;
      }
    }
    if (!has_next) break;
    pb = nb;
    chunk = next_chunk;
    PROF(7)
  }
#ifdef PQD_PROF
  if (lane == 0 && (wave == 0 || wave == 15)) {
    for (int i = 0; i < 12; ++i) prof[(blockIdx.x * 2 + (wave == 15)) * 12 + i] = pt_[i];
  }
#endif
}

// Per device, set on the first launch there: the number of CUs and the dynamic-LDS attribute of the
// two kernels (racing first launches write the same values).
constexpr int DS_MAX_DEVICES = 64;
static int g_cu_count[DS_MAX_DEVICES];
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

hipError_t launch_dict_pages_sliced(pqd_page_result* results, const void* pool, uint32_t pool_chunks, const uint32_t* cursor,
                                    uint32_t n_pages, int elem, const uint8_t* plane_area, hipStream_t stream,
                                    hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (n_pages == 0) return hipSuccess;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= DS_MAX_DEVICES) return hipErrorInvalidDevice;
  if (g_cu_count[dev] == 0) {
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dict_pages_sliced<4>), hipFuncAttributeMaxDynamicSharedMemorySize, DS_LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dict_pages_sliced<8>), hipFuncAttributeMaxDynamicSharedMemorySize, DS_LDS_BYTES);
    if (e != hipSuccess) return e;
    __atomic_store_n(&g_cu_count[dev], prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256, __ATOMIC_RELEASE);
  }
  const int cus = __atomic_load_n(&g_cu_count[dev], __ATOMIC_ACQUIRE);
  // one persistent workgroup per CU; each takes a contiguous share of however many chunks the
  // plan kernel made (a count only the device knows)
  const uint32_t grid = n_pages < (uint32_t)cus ? n_pages : (uint32_t)cus;
  const RleSeg* sp = reinterpret_cast<const RleSeg*>(pool);
#ifdef PQD_PROF
  if (!g_prof && (e = hipMalloc((void**)&g_prof, 1024 * 24 * 8)) != hipSuccess) return e;
#endif
  // Batches made of regular chunks only (counter 3 stays zero) take the path written for them inside the
  // kernel; only the device knows which kind a batch is.  PQD_DICT_REGULAR=0 keeps every batch on the general path.
  const char* reg_env = getenv("PQD_DICT_REGULAR");
  const bool regular = (elem == 4 || plane_area != nullptr) && !(reg_env && strcmp(reg_env, "0") == 0);
  const uint32_t regular_only = reg_env && strcmp(reg_env, "only") == 0 ? 2u : 0u;  // (tests: the general path stays idle)
  const char* stg_env = getenv("PQD_DICT_STAGGER");
  const uint32_t stagger = stg_env ? (uint32_t)atoi(stg_env) : 10u;
#ifndef PQD_PROF
  if (ev_start && ev_stop) {  // timed launch: the dispatch records the kernel's own start and end
    if (elem == 4)
      hipExtLaunchKernelGGL(k_dict_pages_sliced<4>, dim3(grid), dim3(DS_THREADS), DS_LDS_BYTES, stream, ev_start, ev_stop, 0, results, sp,
                            pool_chunks, cursor, n_pages, plane_area, regular ? 1u | regular_only | (stagger << 8) : 0u);
    else
      hipExtLaunchKernelGGL(k_dict_pages_sliced<8>, dim3(grid), dim3(DS_THREADS), DS_LDS_BYTES, stream, ev_start, ev_stop, 0, results, sp,
                            pool_chunks, cursor, n_pages, plane_area, regular ? 1u | regular_only | (stagger << 8) : 0u);
    return hipGetLastError();
  }
#endif
  if (elem == 4)
    hipLaunchKernelGGL(k_dict_pages_sliced<4>, dim3(grid), dim3(DS_THREADS), DS_LDS_BYTES, stream, results, sp, pool_chunks, cursor, n_pages, plane_area, regular ? 1u | regular_only | (stagger << 8) : 0u PROF_ARG);
  else
    hipLaunchKernelGGL(k_dict_pages_sliced<8>, dim3(grid), dim3(DS_THREADS), DS_LDS_BYTES, stream, results, sp, pool_chunks, cursor, n_pages, plane_area, regular ? 1u | regular_only | (stagger << 8) : 0u PROF_ARG);
#ifdef PQD_PROF
  prof_report(grid, stream);
#endif
  return hipGetLastError();
}

}  // namespace pqd
