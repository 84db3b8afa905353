// BINARY (BYTE_ARRAY / FIXED_LEN_BYTE_ARRAY / INT96) dictionary values gathered into one buffer.
//
// The reference hands out Binary objects that are views into the dictionary page
// (parquet-column/.../values/dictionary/PlainValuesDictionary.java:96-113, decodeToBinary :118) and the
// caller copies their bytes (DictionaryValuesReader.readBytes, DictionaryValuesReader.java:76-83, then
// Binary.getBytes / writeTo).  Here a page's values arrive as 8-byte views {offset, length} (the
// dictionary gather with a dictionary of views), and this kernel does the read loop for a whole page:
// offsets[i] = where value i starts, bytes = the values back to back.
//
// One 256-thread workgroup per page, tiles of 2048 values.  A lane owns eight consecutive values: it
// loads their views, a workgroup prefix sum over the lengths gives every value its place, the value
// bytes are fetched with one unaligned 16-byte load each (all eight in flight together), OR-ed into a
// zeroed LDS window at their byte position, and the window goes out in aligned 16-byte stores.
// Values longer than 16 bytes, values at the very end of the dictionary and values that straddle a
// window edge are copied byte by byte.
#include "pqd_device.hpp"
#include "pqd_kernels.hpp"

namespace pqd {

constexpr int BN_THREADS = 256;
constexpr int BN_VPL = 8;  // values per lane and tile  // values per lane and tile
constexpr uint32_t BN_TILE = BN_THREADS * BN_VPL;
constexpr uint32_t BN_WIN = 24 * 1024;  // output bytes staged per window
constexpr uint32_t BN_NONE = 0xFFFFFFFFu;


typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
typedef uint64_t u64_a8 __attribute__((aligned(8)));

// ORs `count` (1..16) bytes held in `v` (little-endian, bytes past count are ignored) into the
// zeroed window at byte position `dst`.
__device__ __forceinline__ void or_bytes(uint32_t* win, uint32_t dst, u32x4 v, uint32_t count) {
  v = keep_leading_bytes(v, (long)count);
  const uint32_t sh = (dst & 3u) * 8u, dw = dst >> 2;
  const uint32_t nd = ((dst & 3u) + count + 3u) >> 2;  // dwords touched: 1..5
  const uint32_t d[5] = {v.x, v.y, v.z, v.w, 0u};
  uint32_t prev = 0u;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const uint32_t piece = (uint32_t)(((((uint64_t)d[k]) << 32) | prev) >> (32u - sh));
    if ((uint32_t)k < nd) atomicOr(&win[dw + k], piece);
    prev = d[k];
  }
}

__global__ __launch_bounds__(BN_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void k_binary_pages(const pqd_binary_page_desc* __restrict__ descs,
                                                             pqd_page_result* results, uint32_t n_pages) {
  __shared__ __attribute__((aligned(16))) uint32_t win[(BN_WIN + 32) / 4];
  __shared__ uint64_t wave_total[BN_THREADS / 64];
  __shared__ uint32_t s_stop, s_stop_status;
  __shared__ uint64_t s_stop_bytes;
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const pqd_binary_page_desc d = descs[page];
  const uint32_t n = d.num_values;
  const uint64_t cap = d.bytes_capacity > 0x7fffffffu ? 0x7fffffffull : d.bytes_capacity;  // int32 offsets
  const PQD_GLOBAL uint64_t* views = (const PQD_GLOBAL uint64_t*)d.views;
  const PQD_GLOBAL uint8_t* dict = (const PQD_GLOBAL uint8_t*)d.dict_bytes;
  PQD_GLOBAL uint32_t* offsets = (PQD_GLOBAL uint32_t*)d.offsets;
  PQD_GLOBAL uint8_t* bytes = (PQD_GLOBAL uint8_t*)d.bytes;
  const PQD_GLOBAL uint8_t* always_there = (const PQD_GLOBAL uint8_t*)(results + page);
  const uint64_t dict_len = d.dict_bytes_len;
  uint32_t running = 0;  // bytes written so far
  uint32_t status = PQD_OK, written = n;
  bool stopped = false;
  // the window is all zero whenever a tile starts: the flush clears what it has read
  for (uint32_t i = tid; i < (BN_WIN + 32) / 16; i += BN_THREADS) ((u32x4*)win)[i] = u32x4{0u, 0u, 0u, 0u};
  if (tid == 0) s_stop = BN_NONE;


  for (uint32_t t0 = 0; t0 < n && !stopped; t0 += BN_TILE) {
    const uint32_t g0 = t0 + tid * BN_VPL;
    uint32_t off[BN_VPL], len[BN_VPL];  // off: only until the loads below are issued
    uint32_t bad_mask = 0, fast_mask = 0;
    {
      uint64_t v[BN_VPL];
#pragma unroll
      for (int j = 0; j < BN_VPL; ++j) v[j] = views[g0 + j < n ? g0 + j : (n - 1)];
#pragma unroll
      for (int j = 0; j < BN_VPL; ++j) {
        off[j] = (uint32_t)v[j];
        len[j] = g0 + j < n ? (uint32_t)(v[j] >> 32) : 0u;
        if (len[j] == 0xFFFFFFFFu || (uint64_t)off[j] + len[j] > dict_len) {
          bad_mask |= 1u << j;
          len[j] = 0;
        }
      }
    }
    // the value bytes: one 16-byte load per value where that is enough and stays inside the
    // dictionary; every lane loads (the others from 16 bytes that are always there: the page's own
    // result record), so that the loads are in flight together
    u32x4 data[BN_VPL];
#pragma unroll
    for (int j = 0; j < BN_VPL; ++j) {
      const bool fast = len[j] != 0 && len[j] <= 16u && (uint64_t)off[j] + 16u <= dict_len;
      if (fast) fast_mask |= 1u << j;
      const PQD_GLOBAL uint8_t* src = fast ? dict + off[j] : always_there;
      const u32x4_a1 raw = *(const PQD_GLOBAL u32x4_a1*)src;
      data[j] = u32x4{raw.x, raw.y, raw.z, raw.w};
    }
    // prefix sums of the lengths: within the lane, the wave, the workgroup (64-bit: a page may ask
    // for more than 4 GiB before it is found not to fit)
    uint64_t lane_total = 0;
#pragma unroll
    for (int j = 0; j < BN_VPL; ++j) lane_total += len[j];
    const uint64_t inc = wave_scan_inclusive(lane_total);
    if (lane == 63) wave_total[wave] = inc;
    __syncthreads();
    uint64_t lane_base = inc - lane_total, tile_total = 0;
#pragma unroll
    for (int k = 0; k < BN_THREADS / 64; ++k) {
      const uint64_t wt = wave_total[k];
      if ((uint32_t)k < wave) lane_base += wt;
      tile_total += wt;
    }
    // the first value that cannot be delivered: a broken view, or bytes beyond the capacity
    if (bad_mask != 0 || (uint64_t)running + lane_base + lane_total > cap) {
      uint64_t at = (uint64_t)running + lane_base;
      uint32_t my_stop = BN_NONE;
#pragma unroll
      for (int j = 0; j < BN_VPL; ++j) {
        const bool fails = ((bad_mask >> j) & 1u) || at + len[j] > cap;
        if (my_stop == BN_NONE && g0 + j < n && fails) my_stop = tid * BN_VPL + j;
        at += len[j];
      }
      if (my_stop != BN_NONE) atomicMin(&s_stop, my_stop);
    }
    __syncthreads();
    const uint32_t stop = s_stop;  // tile-relative index, BN_NONE if the whole tile is fine
    uint32_t pre[BN_VPL];          // wraps only behind the value that stops the page
    {
      uint32_t acc = (uint32_t)lane_base;
#pragma unroll
      for (int j = 0; j < BN_VPL; ++j) {
        pre[j] = acc;
        acc += len[j];
      }
    }
    uint32_t tile_bytes = (uint32_t)tile_total;  // below 2^31 when nothing stops the page
    if (stop != BN_NONE) {  // rare: the page ends here
      if (stop / BN_VPL == tid) {
        uint64_t at = lane_base;
#pragma unroll
        for (int j = 0; j < BN_VPL; ++j) {
          if ((uint32_t)j == stop % BN_VPL) {
            s_stop_bytes = at;
            s_stop_status = ((bad_mask >> j) & 1u) ? PQD_DECODE : PQD_BAD_ARG;
          }
          at += len[j];
        }
      }
      __syncthreads();
      tile_bytes = (uint32_t)s_stop_bytes;
      status = s_stop_status;
      written = t0 + stop;
      stopped = true;
    }
    // where the values start (the value that stops the page gets the end of the one before it)
    if (g0 + BN_VPL <= n && (stop == BN_NONE || tid * BN_VPL + BN_VPL - 1 <= stop)) {
#pragma unroll
      for (int j = 0; j < BN_VPL; j += 4) {
        const u32x4_a4 o4 = {running + pre[j], running + pre[j + 1], running + pre[j + 2], running + pre[j + 3]};
        PQD_STREAM_STORE((PQD_GLOBAL u32x4_a4*)(offsets + g0 + j), o4);
      }
    } else {
#pragma unroll
      for (int j = 0; j < BN_VPL; ++j)
        if (g0 + j < n && tid * BN_VPL + j <= stop) offsets[g0 + j] = running + pre[j];
    }
    // the bytes, one LDS window after the other
    for (uint32_t wb = 0; wb < tile_bytes; wb += BN_WIN) {
      const uint32_t wlen = tile_bytes - wb < BN_WIN ? tile_bytes - wb : BN_WIN;
      PQD_GLOBAL uint8_t* gdst = bytes + running + wb;
      const uint32_t s = (uint32_t)((uintptr_t)gdst & 15u);
      const uint32_t pieces = (s + wlen + 15u) >> 4;
      uint32_t slow_mask = 0;
#pragma unroll
      for (int j = 0; j < BN_VPL; ++j) {
        if (tid * BN_VPL + j >= stop || len[j] == 0) continue;
        const uint32_t b0 = pre[j], b1 = b0 + len[j];
        if (b1 <= wb || b0 >= wb + wlen) continue;
        if (((fast_mask >> j) & 1u) && b0 >= wb && b1 <= wb + wlen) {
          or_bytes(win, s + (b0 - wb), data[j], len[j]);
        } else {
          slow_mask |= 1u << j;
        }
      }
      while (slow_mask) {  // byte by byte: long values, the dictionary's last bytes, window edges
        const uint32_t j = (uint32_t)__builtin_ctz(slow_mask);
        slow_mask &= slow_mask - 1u;
        uint32_t b0 = 0, l = 0;
#pragma unroll
        for (int k = 0; k < BN_VPL; ++k) {
          b0 = (uint32_t)k == j ? pre[k] : b0;
          l = (uint32_t)k == j ? len[k] : l;
        }
        const uint32_t src0 = (uint32_t)views[g0 + j];
        const uint32_t lo = b0 > wb ? b0 : wb, hi = b0 + l < wb + wlen ? b0 + l : wb + wlen;
        for (uint32_t b = lo; b < hi; ++b) {
          const uint32_t dst = s + (b - wb);
          atomicOr(&win[dst >> 2], (uint32_t)dict[(uint64_t)src0 + (b - b0)] << ((dst & 3u) * 8u));
        }
      }
      __syncthreads();
      const uint32_t end = s + wlen;
      for (uint32_t i = tid; i < pieces; i += BN_THREADS) {
        const u32x4 v = ((const u32x4*)win)[i];
        ((u32x4*)win)[i] = u32x4{0u, 0u, 0u, 0u};
        const uint32_t p0 = i * 16u;
        if (p0 >= s && p0 + 16u <= end) {
          PQD_STREAM_STORE((PQD_GLOBAL u32x4*)(gdst - s + p0), v);
        } else {  // the first and the last piece of a window: only the bytes that belong to it
#pragma unroll 1
          for (uint32_t k = 0; k < 16u; ++k) {
            const uint32_t word = k < 8u ? (k < 4u ? v.x : v.y) : (k < 12u ? v.z : v.w);
            if (p0 + k >= s && p0 + k < end) gdst[(long)(p0 + k) - (long)s] = (uint8_t)(word >> ((k & 3u) * 8u));
          }
        }
      }
      __syncthreads();
    }
    running += tile_bytes;
  }
  if (tid == 0) {
    if (!stopped) offsets[n] = running;
    results[page].status = (int32_t)status;
    results[page].values_written = written;
    results[page].bytes_consumed = running;
    results[page].reserved = 0;
  }
}

hipError_t launch_binary_pages(const pqd_binary_page_desc* descs, pqd_page_result* results, uint32_t n_pages,
                               hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  hipLaunchKernelGGL(k_binary_pages, dim3(n_pages), dim3(BN_THREADS), 0, stream, descs, results, n_pages);
  return hipGetLastError();
}

}  // namespace pqd
