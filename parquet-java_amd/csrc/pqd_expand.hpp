// Expansion of one tile of a chunk of the run plan (pqd_rle_plan.hpp) into values, shared by the kernels that
// expand chunks: k_rle_expand (plain hybrid pages: the values are the output) and the direct dictionary chunks
// handled inside k_rle_pages (the values are ids, looked up in the dictionary in memory).
#pragma once

#include "pqd_device.hpp"
#include "pqd_rle_plan.hpp"
#include "../../include/pqd.h"

namespace pqd {

__device__ __forceinline__ uint32_t ex_uni(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }

constexpr int EX_THREADS = 256;
constexpr int EX_UNITS = 4;                                  // units per lane per tile
constexpr uint32_t EX_TILE = EX_THREADS * EX_UNITS * 4;      // 4096 values
static_assert(EX_THREADS * 4 % SEG_BLOCK == 0 && SEG_BLOCK == 256, "a wave's unit row is one table block");



struct ExHead {
  const PQD_GLOBAL uint32_t* words;  // `data` rounded down to a dword
  uint32_t in_off;                   // data & 3
  uint32_t last_word;                // last dword holding a byte of the page
  uint32_t full_words;               // dwords lying wholly before the end of the stream
  uint32_t tail_mask;                // valid bytes of dword `full_words` (0: none)
  uint32_t w;
};

__device__ __forceinline__ uint32_t ex_guarded_word(const ExHead& h, uint32_t i) {
  if (i < h.full_words) return h.words[i];
  if (i == h.full_words && h.tail_mask) return h.words[i] & h.tail_mask;
  return 0u;  // at or past the end of the stream
}

// Value j of a unit whose 64 aligned bits are {hi, lo}; w <= 16 and j are wave-uniform.
__device__ __forceinline__ uint32_t ex_unit_value(uint32_t lo, uint32_t hi, uint32_t w, uint32_t j) {
  const uint32_t off = j * w;
  if (off + w <= 32) return __builtin_amdgcn_ubfe(lo, off, w);
  if (off >= 32) return __builtin_amdgcn_ubfe(hi, off - 32, w);
  return funnel_shr(lo, hi, off) & ((1u << w) - 1u);
}

constexpr uint32_t EK_NONE = 0, EK_PACKED = 1, EK_RLE = 2, EK_SLOW = 3;

// Where the values of a dictionary page's chunk go (k_dict_expand): the expanded ids are looked up in the
// dictionary in memory, value = dictionary.decodeToX(id) (DictionaryValuesReader.java:75-118).  An id
// beyond the dictionary is noted in the page's result (the first such value ends what the page delivers,
// ColumnReaderBase.java:582-616) and entry 0 is read instead.
struct DictSink {
  const PQD_GLOBAL uint8_t* dict;
  uint32_t dict_len;
  PQD_GLOBAL uint8_t* out;  // element 0 of the page's output
  pqd_page_result* result;
};
template <int ELEM>
__device__ __forceinline__ uint32_t sink_check(const DictSink& ds, uint32_t id, uint32_t g) {
  if (id < ds.dict_len) return id;
  atomicMin(&ds.result->values_written, g);
  ds.result->status = PQD_DICT_OOB;
  return 0u;
}
template <int ELEM>
__device__ __forceinline__ void sink_one(const DictSink& ds, uint32_t g, uint32_t id) {
  id = sink_check<ELEM>(ds, id, g);
  if (ELEM == 4) {
    ((PQD_GLOBAL uint32_t*)ds.out)[g] = ((const PQD_GLOBAL uint32_t*)ds.dict)[id];
  } else {
    ((PQD_GLOBAL u32x2_a4*)ds.out)[g] = ((const PQD_GLOBAL u32x2_a4*)ds.dict)[id];
  }
}

// One tile of EX_TILE values starting at output index t0 (a multiple of SEG_BLOCK past v0).
// WIDE: ids of 17..32 bits (five dwords per unit) instead of up to 16 bits (three).
// Units that need the value-by-value treatment (run boundaries off multiples of four, truncated
// runs, the page's last dwords) are not done by their lane: they are put on a list in LDS and the
// whole workgroup then does them one VALUE per lane, so that a wave with one such unit among its 64
// does not walk all its lanes through that path.
// DICT: 0 = the values are the output; 4 / 8 = they are ids into a dictionary of 4- / 8-byte entries (`ds`).
template <bool WIDE, int DICT = 0>
__device__ __forceinline__ void expand_tile(const PQD_LDS RleSeg* cs, const ExHead& h, uint32_t v0, uint32_t t0,
                                            uint32_t v1, PQD_GLOBAL uint32_t* out, uint32_t tid, uint32_t wave,
                                            uint32_t* slow_list, uint32_t* slow_n, uint32_t parity, const DictSink& ds = DictSink()) {
  constexpr int NW = WIDE ? 5 : 3;
  const PQD_LDS uint8_t* table = (const PQD_LDS uint8_t*)(cs + SEG_TABLE_SLOT);
  const uint32_t v1x4 = v1 << 2;
  const bool can_load = h.last_word + 1 >= (uint32_t)NW;  // the page holds at least one whole unit load
  const uint32_t max_wi = can_load ? h.last_word + 1 - NW : 0u;
  uint32_t wi[EX_UNITS], meta[EX_UNITS], other[EX_UNITS];
  // Every step below is done for all units before the next one, so that a step costs one LDS
  // round trip for the lane, not one per unit.
  uint32_t sg[EX_UNITS];
#pragma unroll
  for (int k = 0; k < EX_UNITS; ++k) sg[k] = table[((t0 - v0) >> 8) + (uint32_t)k * (EX_THREADS * 4 / SEG_BLOCK) + wave];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    uint32_t nx[EX_UNITS];
#pragma unroll
    for (int k = 0; k < EX_UNITS; ++k) nx[k] = cs[sg[k] + 1].start4;
#pragma unroll
    for (int k = 0; k < EX_UNITS; ++k) {
      const uint32_t g0 = t0 + 4 * ((uint32_t)k * EX_THREADS + tid);
      sg[k] += nx[k] <= g0 * 4 + 3 ? 1u : 0u;  // start4 <= g0*4+3 <=> start <= g0; valid lanes stop at the sentinel
    }
  }
  for (;;) {  // rare: more than two segment boundaries inside 256 values
    bool any = false;
    uint32_t nx[EX_UNITS];
#pragma unroll
    for (int k = 0; k < EX_UNITS; ++k) nx[k] = cs[sg[k] + 1].start4;
#pragma unroll
    for (int k = 0; k < EX_UNITS; ++k) {
      const uint32_t g0 = t0 + 4 * ((uint32_t)k * EX_THREADS + tid);
      const bool adv = g0 < v1 && nx[k] <= g0 * 4 + 3;
      sg[k] += adv ? 1u : 0u;
      any |= adv;
    }
    if (!__builtin_amdgcn_ballot_w64(any)) break;
  }
  uint32_t sk[EX_UNITS], e4s[EX_UNITS];
#pragma unroll
  for (int k = 0; k < EX_UNITS; ++k) {
    sk[k] = cs[sg[k]].start4;
    other[k] = cs[sg[k]].src;  // stream offset of a bit-packed segment / the repeated value of an RLE run
    e4s[k] = cs[sg[k] + 1].start4;
  }
#pragma unroll
  for (int k = 0; k < EX_UNITS; ++k) {
    const uint32_t g0 = t0 + 4 * ((uint32_t)k * EX_THREADS + tid);
    const uint32_t lim4 = e4s[k] < v1x4 ? e4s[k] : v1x4;
    const bool is_packed = (sk[k] & SEG_PACKED_BIT) != 0;
    const bool whole = g0 * 4 + 16 <= lim4 && !(sk[k] & SEG_PARTIAL_BIT);
    // a segment holds at most PLAN_SEG_MAX_BYTES of payload: the bit offset inside it fits 32 bits
    const uint32_t byte0 = h.in_off + other[k];
    const uint32_t bitrel = (byte0 & 3u) * 8 + (g0 - (sk[k] >> 2)) * h.w;
    const uint32_t w_i = (byte0 >> 2) + (bitrel >> 5);
    const bool fast = whole && is_packed && can_load && w_i <= max_wi;
    const uint32_t kind = g0 >= v1 ? EK_NONE : (is_packed ? (fast ? EK_PACKED : EK_SLOW) : (whole ? EK_RLE : EK_SLOW));
    wi[k] = kind == EK_PACKED ? w_i : 0u;
    meta[k] = (bitrel & 31u) | (kind << 8) | (sg[k] << 16);
  }
  // every lane issues every load (units with nothing to read take the page's first dwords, or
  // nothing at all when the page is shorter than one unit load)
  uint32_t d[EX_UNITS][NW];
#pragma unroll
  for (int k = 0; k < EX_UNITS; ++k) {
    if (can_load) {  // uniform
      if (WIDE) {
        const u32x4_a4 a = *(const PQD_GLOBAL u32x4_a4*)(h.words + wi[k]);
        d[k][0] = a.x;
        d[k][1] = a.y;
        d[k][2] = a.z;
        d[k][WIDE ? 3 : 0] = a.w;
        d[k][WIDE ? 4 : 0] = h.words[wi[k] + 4];
      } else {
        const u32x3_a4 a = *(const PQD_GLOBAL u32x3_a4*)(h.words + wi[k]);
        d[k][0] = a.x;
        d[k][1] = a.y;
        d[k][2] = a.z;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NW; ++j) d[k][j] = 0;
    }
  }
  const uint32_t mask = mask32(h.w);
  uint32_t ids[DICT ? EX_UNITS : 1][4];  // dictionary pages: the units' ids, looked up together below
#pragma unroll
  for (int k = 0; k < EX_UNITS; ++k) {
    const uint32_t g0 = t0 + 4 * ((uint32_t)k * EX_THREADS + tid);
    const uint32_t kind = (meta[k] >> 8) & 3u;
    const uint32_t b = meta[k] & 31u;
    uint32_t a0, a1, a2, a3;
    if (WIDE) {
      Window5 wnd = {d[k][0], d[k][1], d[k][2], d[k][WIDE ? 3 : 0], d[k][WIDE ? 4 : 0]};
      wnd.shr(b);
      a0 = wnd.d0 & mask;
      wnd.drop(h.w);
      a1 = wnd.d0 & mask;
      wnd.drop(h.w);
      a2 = wnd.d0 & mask;
      wnd.drop(h.w);
      a3 = wnd.d0 & mask;
    } else {
      const uint32_t lo = funnel_shr(d[k][0], d[k][1], b), hi = funnel_shr(d[k][1], d[k][2], b);
      a0 = ex_unit_value(lo, hi, h.w, 0);
      a1 = ex_unit_value(lo, hi, h.w, 1);
      a2 = ex_unit_value(lo, hi, h.w, 2);
      a3 = ex_unit_value(lo, hi, h.w, 3);
    }
    if (kind != EK_PACKED) a0 = a1 = a2 = a3 = other[k];
    if (DICT) {
      ids[DICT ? k : 0][0] = a0;
      ids[DICT ? k : 0][1] = a1;
      ids[DICT ? k : 0][2] = a2;
      ids[DICT ? k : 0][3] = a3;
    }
    if (kind == EK_SLOW) {
      const uint32_t idx = atomicAdd(&slow_n[parity], 1u);
      slow_list[2 * idx] = g0;
      slow_list[2 * idx + 1] = meta[k] >> 16;  // the segment of the unit's first value
    } else if (DICT) {
      // (looked up below, all units of the lane at once)
    } else if (g0 + 4 <= v1) {
      const u32x4_a4 r4 = {a0, a1, a2, a3};
      *(PQD_GLOBAL u32x4_a4*)(out + g0) = r4;
    } else if (g0 < v1) {
      out[g0] = a0;
      if (g0 + 1 < v1) out[g0 + 1] = a1;
      if (g0 + 2 < v1) out[g0 + 2] = a2;
    }
  }
  if constexpr (DICT != 0) {
    // the gathers of all units are issued before the first store (a store between them would make the
    // compiler wait for the loads before it)
    uint32_t lo[EX_UNITS][4], hi[DICT == 8 ? EX_UNITS : 1][4];
#pragma unroll
    for (int k = 0; k < EX_UNITS; ++k) {
      const uint32_t g0 = t0 + 4 * ((uint32_t)k * EX_THREADS + tid);
      const uint32_t kind = (meta[k] >> 8) & 3u;
      const bool mine = kind == EK_PACKED || kind == EK_RLE;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t id = (mine && g0 + j < v1) ? sink_check<DICT>(ds, ids[k][j], g0 + j) : 0u;
        if (DICT == 4) {
          lo[k][j] = ((const PQD_GLOBAL uint32_t*)ds.dict)[id];
        } else {
          const u32x2_a4 e = ((const PQD_GLOBAL u32x2_a4*)ds.dict)[id];
          lo[k][j] = e.x;
          hi[DICT == 8 ? k : 0][j] = e.y;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < EX_UNITS; ++k) {
      const uint32_t g0 = t0 + 4 * ((uint32_t)k * EX_THREADS + tid);
      const uint32_t kind = (meta[k] >> 8) & 3u;
      if ((kind != EK_PACKED && kind != EK_RLE) || g0 >= v1) continue;
      if (DICT == 4) {
        PQD_GLOBAL uint32_t* o = (PQD_GLOBAL uint32_t*)ds.out + g0;
        if (g0 + 4 <= v1) {
          const u32x4_a4 r4 = {lo[k][0], lo[k][1], lo[k][2], lo[k][3]};
          *(PQD_GLOBAL u32x4_a4*)o = r4;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (g0 + j < v1) o[j] = lo[k][j];
        }
      } else {
        PQD_GLOBAL u32x2_a4* o = (PQD_GLOBAL u32x2_a4*)ds.out + g0;
        if (g0 + 4 <= v1) {
          const u32x4_a4 r0 = {lo[k][0], hi[DICT == 8 ? k : 0][0], lo[k][1], hi[DICT == 8 ? k : 0][1]};
          const u32x4_a4 r1 = {lo[k][2], hi[DICT == 8 ? k : 0][2], lo[k][3], hi[DICT == 8 ? k : 0][3]};
          *(PQD_GLOBAL u32x4_a4*)o = r0;
          *(PQD_GLOBAL u32x4_a4*)(o + 2) = r1;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (g0 + j < v1) {
              const u32x2_a4 x = {lo[k][j], hi[DICT == 8 ? k : 0][j]};
              o[j] = x;
            }
        }
      }
    }
  }
  __syncthreads();
  const uint32_t n_slow = slow_n[parity];
  if (tid == 0) slow_n[parity ^ 1u] = 0;  // the next tile's counter (this one is read until the barrier below)
  for (uint32_t i = tid; i < 4 * n_slow; i += EX_THREADS) {
    const uint32_t g = slow_list[2 * (i >> 2)] + (i & 3u);
    uint32_t t = slow_list[2 * (i >> 2) + 1];
    if (g >= v1) continue;
    while (cs[t + 1].start4 <= g * 4 + 3) ++t;  // stops at the sentinel (start == v1 > g)
    const uint32_t tk = cs[t].start4, src = cs[t].src;
    uint32_t v = src;  // the repeated value of an RLE run
    if (tk & SEG_PACKED_BIT) {
      const uint32_t byte0 = h.in_off + src;
      const uint32_t bitrel = (byte0 & 3u) * 8 + (g - (tk >> 2)) * h.w;
      const uint32_t i0 = (byte0 >> 2) + (bitrel >> 5), i1 = i0 + 1;
      // clamped addresses: never past the page; bytes at or past the end of the stream read as zero
      const uint32_t x0 = h.words[i0 < h.last_word ? i0 : h.last_word], x1 = h.words[i1 < h.last_word ? i1 : h.last_word];
      const uint32_t y0 = i0 < h.full_words ? x0 : (i0 == h.full_words ? x0 & h.tail_mask : 0u);
      const uint32_t y1 = i1 < h.full_words ? x1 : (i1 == h.full_words ? x1 & h.tail_mask : 0u);
      v = funnel_shr(y0, y1, bitrel & 31u) & mask;
    }
    if (DICT)
      sink_one<DICT ? DICT : 4>(ds, g, v);
    else
      out[g] = v;
  }
  __syncthreads();
}

// The short chunks of dictionary pages (DS_DIRECT_MAX, pqd_rle_plan.hpp): expanded like plain hybrid chunks, every id
// looked up in the dictionary in memory.  They are the last PLAN_CTR_DIRECT chunks of the pool; the workgroups of the
// calling kernel (EX_THREADS threads each) stride over them.  cs_buf: SEG_CHUNK segments; slow_list: 2 * EX_THREADS *
// EX_UNITS words; slow_n: 2 words (all in LDS).
template <int ELEM>
__device__ __forceinline__ void dict_expand_direct_chunks(pqd_page_result* __restrict__ results, const RleSeg* __restrict__ pool,
                                                          uint32_t pool_chunks, uint32_t n_pages, const uint32_t* __restrict__ cursor,
                                                          RleSeg* cs_buf, uint32_t* slow_list, uint32_t* slow_n) {
  const uint32_t n_direct = ex_uni(cursor[PLAN_CTR_DIRECT]);
  // the few irregular chunks of a batch that took the regular path of the sliced gather (listed by the plan kernel)
  const bool regular_mode = ex_uni(cursor[PLAN_CTR_IRREGULAR]) == 0u &&
                            (ex_uni(cursor[PLAN_CTR_RESIDENT]) == 0u || ex_uni(cursor[PLAN_CTR_SWEPT]) == 0u);  // as k_dict_pages_sliced decides
  uint32_t n_listed = regular_mode ? ex_uni(cursor[PLAN_CTR_IRREGULAR_LISTED]) : 0u;
  if (n_listed > PLAN_IRREGULAR_LIST) n_listed = PLAN_IRREGULAR_LIST;
  if (n_direct + n_listed == 0u) return;
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = ex_uni(tid >> 6);
  const PQD_LDS RleSeg* cs = (const PQD_LDS RleSeg*)cs_buf;
  uint32_t parity = 0;
  __syncthreads();  // whatever used this LDS before is done
  if (tid < 2) slow_n[tid] = 0;
  const uint32_t back_cap = pool_chunks - plan_front_limit(pool_chunks, n_pages, true);  // chunks the plan kernel could grant at the end
  const uint32_t n_mine = n_direct < back_cap ? n_direct : back_cap;
  for (uint32_t i = blockIdx.x; i < n_mine + n_listed; i += gridDim.x) {
    const uint32_t c = i < n_mine ? pool_chunks - 1u - i : ex_uni(cursor[PLAN_CTR_LIST0 + (i - n_mine)]);
    if (c >= pool_chunks) continue;
    __syncthreads();  // everybody is done with the previous chunk
    {
      const PQD_GLOBAL uint64_t* src = (const PQD_GLOBAL uint64_t*)(pool + (size_t)c * SEG_CHUNK);
      ((PQD_LDS uint64_t*)cs_buf)[tid] = src[tid];  // SEG_CHUNK == EX_THREADS slots of 8 bytes
    }
    __syncthreads();
    const PQD_LDS RleChunkHead* hd = (const PQD_LDS RleChunkHead*)(cs + SEG_HEAD_SLOT);
    const uint32_t dlo = ex_uni((uint32_t)hd->data), dhi = ex_uni((uint32_t)(hd->data >> 32));
    const uint64_t data = ((uint64_t)dhi << 32) | dlo;
    const uint32_t v0 = ex_uni(hd->v0), v1 = ex_uni(hd->v1);
    const uint32_t data_len = ex_uni(hd->data_len), stream_end = ex_uni(hd->stream_end);
    ExHead h;
    h.in_off = dlo & 3u;
    h.words = (const PQD_GLOBAL uint32_t*)(data - h.in_off);
    h.last_word = data_len ? (h.in_off + data_len - 1) >> 2 : 0u;
    const uint32_t end_bytes = h.in_off + stream_end;
    h.full_words = end_bytes >> 2;
    h.tail_mask = (end_bytes & 3u) ? ((1u << (8 * (end_bytes & 3u))) - 1u) : 0u;
    h.w = ex_uni(hd->w);
    DictSink ds;
    ds.dict = (const PQD_GLOBAL uint8_t*)(((uint64_t)ex_uni((uint32_t)(hd->dict >> 32)) << 32) | ex_uni((uint32_t)hd->dict));
    ds.dict_len = ex_uni(hd->dict_len);
    ds.out = (PQD_GLOBAL uint8_t*)(((uint64_t)ex_uni((uint32_t)(hd->out >> 32)) << 32) | ex_uni((uint32_t)hd->out));
    ds.result = results + ex_uni(hd->page);
    for (uint32_t t0 = v0; t0 < v1; t0 += EX_TILE, parity ^= 1u)
      expand_tile<false, ELEM>(cs, h, v0, t0, v1, nullptr, tid, wave, slow_list, slow_n, parity, ds);
  }
}

}  // namespace pqd
