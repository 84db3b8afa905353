// Decoded levels -> columnar structure: validity bitmap and value positions from definition levels,
// null scatter of the page's dense values, list offsets from repetition levels.
//
// The reference assembles records one value at a time: ColumnReaderBase.checkRead reads one repetition
// and one definition level per value, a value is present exactly when dl == maxDefinitionLevel and a
// record starts where rl == 0 (parquet-column/.../impl/ColumnReaderBase.java:642-668, isNull :322-325;
// the per-record state machine is parquet-column/.../io/RecordReaderImplementation.java:409-429).  A
// columnar consumer wants the same facts as arrays.  One 256-thread workgroup per page, tiles of 4096
// levels (16 per lane), a running carry between tiles; everything is a prefix count over level
// predicates (DPP wave scans + one LDS hop between the four waves).
// HBM-bound: 4 bytes per level in, 1/8 byte (+ 4 with positions) out.
#include "pqd_device.hpp"
#include "pqd_kernels.hpp"

namespace pqd {

constexpr int LV_THREADS = 256;
constexpr int LV_PER_LANE = 16;
constexpr uint32_t LV_TILE = LV_THREADS * LV_PER_LANE;

// Exclusive prefix of `v` over the workgroup and the workgroup's total (all lanes call it).
__device__ __forceinline__ uint32_t block_exclusive(uint32_t v, uint32_t* wave_totals, uint32_t& total) {
  const uint32_t inc = wave_scan_inclusive(v);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  __syncthreads();  // wave_totals may still be read from the previous call
  if (lane == 63) wave_totals[wave] = inc;
  __syncthreads();
  uint32_t before = 0, sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < LV_THREADS / 64; ++k) {
    const uint32_t t = wave_totals[k];
    before += k < wave ? t : 0u;
    sum += t;
  }
  total = sum;
  return before + inc - v;
}

// 16 levels of a lane (zero beyond n)
__device__ __forceinline__ void load_levels(const int32_t* levels, uint32_t first, uint32_t n, int32_t (&lv)[LV_PER_LANE]) {
  if (first + LV_PER_LANE <= n) {
#pragma unroll
    for (int q = 0; q < LV_PER_LANE / 4; ++q) {
      const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(levels + first + 4 * q);
      lv[4 * q] = (int32_t)v.x;
      lv[4 * q + 1] = (int32_t)v.y;
      lv[4 * q + 2] = (int32_t)v.z;
      lv[4 * q + 3] = (int32_t)v.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < LV_PER_LANE; ++j) lv[j] = first + j < n ? levels[first + j] : 0;
  }
}

// 64-bit ballot of a wave as two dwords in every lane's registers is not needed: the ballots below
// are wave-uniform values (scalar registers).
__device__ __forceinline__ uint32_t lanes_below(uint64_t bits, uint32_t lane) {
  return (uint32_t)__builtin_popcountll(bits & ((1ull << lane) - 1ull));
}

// Exclusive prefix of a wave-uniform count over the workgroup's four waves, and the total.
__device__ __forceinline__ uint32_t waves_exclusive(uint32_t wave_count, uint32_t* wave_totals, uint32_t& total) {
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  __syncthreads();  // wave_totals may still be read from the previous call
  if (lane == 0) wave_totals[wave] = wave_count;
  __syncthreads();
  uint32_t before = 0, sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < LV_THREADS / 64; ++k) {
    const uint32_t t = wave_totals[k];
    before += k < wave ? t : 0u;
    sum += t;
  }
  total = sum;
  return before;
}

// The bitmap alone (no value positions wanted): a lane takes sixteen consecutive levels with four 16-byte
// loads and writes their sixteen bits as two bytes.
__device__ __forceinline__ void validity_bitmap_only(const pqd_validity_desc& d, pqd_page_result* results, uint32_t page,
                                                     uint32_t* wave_totals) {
  const uint32_t n = d.n_levels;
  uint32_t carry = 0;  // non-null values before the tile
  for (uint32_t t0 = 0; t0 < n; t0 += LV_TILE) {
    const uint32_t first = t0 + threadIdx.x * LV_PER_LANE;
    uint32_t bits = 0;
    if (d.def_levels) {
      int32_t lv[LV_PER_LANE];
      load_levels(d.def_levels, first, n, lv);
#pragma unroll
      for (int j = 0; j < LV_PER_LANE; ++j) bits |= (first + j < n && lv[j] == d.max_def) ? 1u << j : 0u;
    } else {  // a required column: every value is present
      const uint32_t left = first < n ? n - first : 0u;
      bits = left >= LV_PER_LANE ? 0xFFFFu : (1u << left) - 1u;
    }
    uint32_t tile_total;
    (void)block_exclusive((uint32_t)__builtin_popcount(bits), wave_totals, tile_total);
    if (first < n) {
      if (d.validity) {
        const uint16_t b16 = (uint16_t)bits;  // padding bits of the last byte are zero
        const uint32_t left_bytes = (n - first + 7) >> 3;
        uint8_t* o = d.validity + (first >> 3);
        if (left_bytes >= 2)
          __builtin_memcpy(o, &b16, 2);
        else
          o[0] = (uint8_t)b16;
      }
    }
    carry += tile_total;
  }
  if (threadIdx.x == 0) {
    pqd_page_result r = {PQD_OK, n, 0u, carry};
    results[page] = r;
  }
}

// A tile is 4096 levels; a wave takes 1024 of them as sixteen rows of 64, one level per lane: loads and
// stores of a row are contiguous, a row's presence bits are one ballot, a value's position is the count
// of the rows before plus the bits of the lanes below.
__global__ __launch_bounds__(LV_THREADS) void k_def_validity(const pqd_validity_desc* __restrict__ descs,
                                                             pqd_page_result* __restrict__ results, uint32_t n_pages) {
  __shared__ uint32_t wave_totals[LV_THREADS / 64];
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_validity_desc d = descs[page];
  if (!d.value_index) {
    validity_bitmap_only(d, results, page, wave_totals);
    return;
  }
  const uint32_t n = d.n_levels;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  uint32_t carry = 0;  // non-null values before the tile
  for (uint32_t t0 = 0; t0 < n; t0 += LV_TILE) {
    const uint32_t w0 = t0 + wave * (LV_TILE / 4);  // this wave's first level
    uint64_t bits[LV_PER_LANE];
    uint32_t count = 0;
    if (d.def_levels) {
      int32_t lv[LV_PER_LANE];
#pragma unroll
      for (int i = 0; i < LV_PER_LANE; ++i) {
        const uint32_t at = w0 + 64u * i + lane;
        lv[i] = at < n ? d.def_levels[at] : d.max_def - 1;
      }
#pragma unroll
      for (int i = 0; i < LV_PER_LANE; ++i) {
        bits[i] = __builtin_amdgcn_ballot_w64(lv[i] == d.max_def && w0 + 64u * i + lane < n);
        count += (uint32_t)__builtin_popcountll(bits[i]);
      }
    } else {  // a required column: every value is present
#pragma unroll
      for (int i = 0; i < LV_PER_LANE; ++i) {
        bits[i] = __builtin_amdgcn_ballot_w64(w0 + 64u * i + lane < n);
        count += (uint32_t)__builtin_popcountll(bits[i]);
      }
    }
    uint32_t tile_total;
    uint32_t rank0 = carry + waves_exclusive(count, wave_totals, tile_total);
    if (d.validity && w0 < n) {
      // lane i writes row i's eight bytes (padding bits of the page's last byte are zero)
      uint64_t mine = 0;
#pragma unroll
      for (int i = 0; i < LV_PER_LANE; ++i) mine = lane == (uint32_t)i ? bits[i] : mine;
      const uint32_t first = w0 + 64u * lane;
      if (lane < LV_PER_LANE && first < n) {
        uint8_t* o = d.validity + (first >> 3);
        if (first + 64u <= n) {
          __builtin_memcpy(o, &mine, 8);
        } else {
          const uint32_t nb = (n - first + 7u) >> 3;
          for (uint32_t k = 0; k < nb; ++k) o[k] = (uint8_t)(mine >> (8u * k));
        }
      }
    }
    if (d.value_index) {
#pragma unroll
      for (int i = 0; i < LV_PER_LANE; ++i) {
        const uint32_t at = w0 + 64u * i + lane;
        const bool present = (bits[i] >> lane) & 1ull;
        if (at < n) d.value_index[at] = present ? (int32_t)(rank0 + lanes_below(bits[i], lane)) : -1;
        rank0 += (uint32_t)__builtin_popcountll(bits[i]);
      }
    }
    carry += tile_total;
  }
  if (threadIdx.x == 0) {
    pqd_page_result r = {PQD_OK, n, 0u, carry};
    results[page] = r;
  }
}

template <int ELEM>
__global__ __launch_bounds__(LV_THREADS) void k_scatter_nulls(const pqd_scatter_desc* __restrict__ descs,
                                                              pqd_page_result* __restrict__ results, uint32_t n_pages,
                                                              uint32_t elem_size) {
  __shared__ uint32_t wave_totals[LV_THREADS / 64];
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_scatter_desc d = descs[page];
  const uint32_t n = d.n_slots;
  const uint32_t K = ELEM ? (uint32_t)ELEM : elem_size;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  uint32_t carry = 0;
  uint32_t first_missing = ~0u;  // first slot whose value lies beyond n_values (this lane's view)
  for (uint32_t t0 = 0; t0 < n; t0 += LV_TILE) {
    const uint32_t w0 = t0 + wave * (LV_TILE / 4);
    // lane i < 16 fetches row i's 64 validity bits
    uint64_t mine = 0;
    {
      const uint32_t first = w0 + 64u * lane;
      if (lane < LV_PER_LANE && first < n) {
        if (!d.validity) {
          mine = ~0ull;
        } else {
          const uint8_t* v = d.validity + (first >> 3);
          if (first + 64u <= n) {
            __builtin_memcpy(&mine, v, 8);
          } else {
            const uint32_t nb = (n - first + 7u) >> 3;
            for (uint32_t k = 0; k < nb; ++k) mine |= (uint64_t)v[k] << (8u * k);
          }
        }
        if (n - first < 64u) mine &= (1ull << (n - first)) - 1ull;
      }
    }
    const uint32_t my_count = (uint32_t)__builtin_popcountll(mine);
    const uint32_t inc = wave_scan_inclusive(my_count);  // rows are lanes 0..15, the other lanes add nothing
    const uint32_t wave_count = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    uint32_t tile_total;
    const uint32_t wave_rank = carry + waves_exclusive(wave_count, wave_totals, tile_total);
    const uint32_t row_rank = wave_rank + inc - my_count;
#pragma unroll
    for (int i = 0; i < LV_PER_LANE; ++i) {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, i);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), i);
      const uint64_t bits = ((uint64_t)hi << 32) | lo;
      const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)row_rank, i);
      const uint32_t at = w0 + 64u * i + lane;
      if (at >= n) continue;
      const bool valid = (bits >> lane) & 1ull;
      const uint32_t rank = base + lanes_below(bits, lane);
      const bool have = valid && rank < d.n_values;
      if (valid && !have && first_missing == ~0u) first_missing = at;
      if (ELEM == 4) {
        reinterpret_cast<uint32_t*>(d.out)[at] = have ? reinterpret_cast<const uint32_t*>(d.values)[rank] : 0u;
      } else if (ELEM == 8) {
        const u32x2_a4 z = {0u, 0u};
        reinterpret_cast<u32x2_a4*>(d.out)[at] = have ? reinterpret_cast<const u32x2_a4*>(d.values)[rank] : z;
      } else {
        uint8_t* o = reinterpret_cast<uint8_t*>(d.out) + (size_t)at * K;
        const uint8_t* v = reinterpret_cast<const uint8_t*>(d.values) + (size_t)rank * K;
        for (uint32_t b = 0; b < K; ++b) o[b] = have ? v[b] : (uint8_t)0;
      }
    }
    carry += tile_total;
  }
  // the first slot without a value, over the workgroup
  __syncthreads();
  if (threadIdx.x == 0) wave_totals[0] = ~0u;
  __syncthreads();
  if (first_missing != ~0u) atomicMin(&wave_totals[0], first_missing);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t miss = wave_totals[0];
    pqd_page_result r = {miss == ~0u ? PQD_OK : PQD_DECODE, miss == ~0u ? n : miss, 0u, carry};
    results[page] = r;
  }
}

__global__ __launch_bounds__(LV_THREADS) void k_rep_offsets(const pqd_offsets_desc* __restrict__ descs,
                                                            pqd_page_result* __restrict__ results, uint32_t n_pages) {
  __shared__ uint32_t wave_totals[LV_THREADS / 64];
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_offsets_desc d = descs[page];
  const uint32_t n = d.n_levels;
  uint32_t rows = 0, slots = 0;  // before the tile
  for (uint32_t t0 = 0; t0 < n; t0 += LV_TILE) {
    const uint32_t first = t0 + threadIdx.x * LV_PER_LANE;
    int32_t rl[LV_PER_LANE], dl[LV_PER_LANE];
    load_levels(d.rep_levels, first, n, rl);
    load_levels(d.def_levels, first, n, dl);
    uint32_t row_bits = 0, slot_bits = 0;
#pragma unroll
    for (int j = 0; j < LV_PER_LANE; ++j) {
      const bool in = first + j < n;
      row_bits |= (in && rl[j] == 0) ? 1u << j : 0u;
      slot_bits |= (in && dl[j] >= d.slot_def) ? 1u << j : 0u;
    }
    uint32_t tile_rows, tile_slots;
    uint32_t row = rows + block_exclusive((uint32_t)__builtin_popcount(row_bits), wave_totals, tile_rows);
    uint32_t slot = slots + block_exclusive((uint32_t)__builtin_popcount(slot_bits), wave_totals, tile_slots);
#pragma unroll
    for (int j = 0; j < LV_PER_LANE; ++j) {
      if ((row_bits >> j) & 1u) {
        if (row < d.rows_capacity) {
          d.offsets[row] = (int32_t)slot;
          if (d.row_valid) d.row_valid[row] = dl[j] >= d.row_def ? 1 : 0;
        }
        ++row;
      }
      slot += (slot_bits >> j) & 1u;
    }
    rows += tile_rows;
    slots += tile_slots;
  }
  if (threadIdx.x == 0) {
    if (rows <= d.rows_capacity) d.offsets[rows] = (int32_t)slots;
    // more rows than the caller made room for: the offsets of the first rows_capacity rows are written
    pqd_page_result r = {rows <= d.rows_capacity ? PQD_OK : PQD_BAD_ARG, rows, 0u, slots};
    results[page] = r;
  }
}

hipError_t launch_def_validity(const pqd_validity_desc* descs, pqd_page_result* results, uint32_t n_pages, hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  hipLaunchKernelGGL(k_def_validity, dim3(n_pages), dim3(LV_THREADS), 0, stream, descs, results, n_pages);
  return hipGetLastError();
}

hipError_t launch_scatter_nulls(const pqd_scatter_desc* descs, pqd_page_result* results, uint32_t n_pages, int elem,
                                hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  const dim3 grid(n_pages), block(LV_THREADS);
  if (elem == 4)
    hipLaunchKernelGGL(k_scatter_nulls<4>, grid, block, 0, stream, descs, results, n_pages, 4u);
  else if (elem == 8)
    hipLaunchKernelGGL(k_scatter_nulls<8>, grid, block, 0, stream, descs, results, n_pages, 8u);
  else
    hipLaunchKernelGGL(k_scatter_nulls<0>, grid, block, 0, stream, descs, results, n_pages, (uint32_t)elem);
  return hipGetLastError();
}

hipError_t launch_rep_offsets(const pqd_offsets_desc* descs, pqd_page_result* results, uint32_t n_pages, hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  hipLaunchKernelGGL(k_rep_offsets, dim3(n_pages), dim3(LV_THREADS), 0, stream, descs, results, n_pages);
  return hipGetLastError();
}

}  // namespace pqd
