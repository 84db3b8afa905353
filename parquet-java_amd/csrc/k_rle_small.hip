// Small RLE / bit-packing hybrid pages (levels, booleans, dictionary ids as ints) decoded in one go.
//
// Follows RunLengthBitPackingHybridDecoder.readNext (parquet-column/.../values/rle/
// RunLengthBitPackingHybridDecoder.java:80-109) for pages whose stream fits LDS: at most 8 KiB of runs,
// at most 32768 values, bit widths 1..24.  That is every level page of a 20000-value data page, whose
// streams are made of very short runs (an RLE run wherever eight equal levels meet, a few bit-packed
// groups in between: a header every 10 to 20 values), so finding the headers is the work.
// One workgroup of 256 threads per page:
//   1. the stream is copied to LDS;
//   2. every thread reads each of the 64 bytes of its block as if a run header started there and notes
//      where that run would end; back to front it turns these into "a chain that enters my block here
//      leaves it there";
//   3. one thread hops from block to block along the real chain (at most 128 hops, whatever the number
//      of runs) and notes where the chain enters each block;
//   4. every thread lists the runs of its block from that entry; two prefix sums give each run its
//      number and its first output value; the runs go into a table, their first values into a bitmap;
//   5. every thread produces four consecutive values per step: the run of a value is the number of
//      bitmap bits at or before it (a prefix count per 64-bit word plus a popcount), its bits come from
//      the stream in LDS; 16-byte stores, the threads side by side.
// Anything this scheme does not cover (headers of more than four bytes, zero-length runs, runs cut off
// by the end of the stream, pages that end short of their values) leaves the page untouched and marked
// for the kernels that know every rule of the reference (k_rle_plan_chunks / k_rle_pages).
// HBM-bound in steps 1 and 5 (w/8 bytes in, 4 bytes out per value); steps 2-4 are LDS latency.
#include "pqd_device.hpp"
#include "pqd_kernels.hpp"
#include "pqd_rle_plan.hpp"

namespace pqd {

constexpr int RS_THREADS = 256;
constexpr uint32_t RS_BLOCK = 64;  // stream bytes per block (a lane per byte in step 2, a thread per block in step 4)
constexpr uint32_t RS_MAX_BYTES = 8192;
constexpr uint32_t RS_BLOCKS = RS_MAX_BYTES / RS_BLOCK;
static_assert(RS_BLOCKS <= (uint32_t)RS_THREADS, "a thread per block");
constexpr uint32_t RS_GROUP = 8 * RS_BLOCK;  // step 3: blocks are hopped over in groups of eight
constexpr uint32_t RS_GROUPS = RS_MAX_BYTES / RS_GROUP;
static_assert(RS_GROUPS * RS_BLOCK == 4u * RS_THREADS, "four group chains per thread");
constexpr uint32_t RS_MAX_VALUES = 32768;
constexpr uint32_t RS_COMPLEX = 0xFFFFu, RS_END = 0xFFFEu;
constexpr uint32_t RS_BIAS = 1u << 20;  // above first value * w (< 2^15 * 24)
constexpr uint32_t RS_MAX_WIDTH = 24;

struct SmallRun {
  uint32_t len, nvals, nb;
  bool packed, complex;
};

// The run whose header starts with the bytes `lo`; vb = bytes of an RLE run's value.  Headers of up
// to three bytes: counts below 2^20, which covers every run a page of 32768 values needs.
__device__ __forceinline__ SmallRun small_run(uint32_t lo, uint32_t w, uint32_t vb) {
  SmallRun r;
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
__device__ __forceinline__ uint32_t rs_exclusive(uint32_t v, uint32_t* wave_totals, uint32_t& total) {
  const uint32_t inc = wave_scan_inclusive(v);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  __syncthreads();
  if (lane == 63) wave_totals[wave] = inc;
  __syncthreads();
  uint32_t before = 0;
  total = 0;
#pragma unroll
  for (uint32_t k = 0; k < RS_THREADS / 64; ++k) {
    const uint32_t t = wave_totals[k];
    before += k < wave ? t : 0u;
    total += t;
  }
  return before + inc - v;
}

__global__ __launch_bounds__(RS_THREADS) void k_rle_small_pages(const pqd_page_desc* __restrict__ descs,
                                                                pqd_page_result* __restrict__ results,
                                                                RlePagePlan* __restrict__ plans, uint32_t n_pages) {
  __shared__ __attribute__((aligned(16))) uint8_t pg[RS_MAX_BYTES + 96];
  __shared__ __attribute__((aligned(16))) uint16_t tab[RS_MAX_BYTES + 64];  // step 2/3: exits; step 4/5: the run table (as words)
  __shared__ __attribute__((aligned(16))) uint32_t flags[RS_MAX_VALUES / 32 + 2];
  __shared__ uint16_t wbase[RS_MAX_VALUES / 64 + 4];
  __shared__ uint16_t entry_of[RS_BLOCKS];
  __shared__ uint32_t wave_totals[RS_THREADS / 64];
  __shared__ uint32_t s_ok, s_consumed;
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_page_desc d = descs[page];
  const uint32_t tid = threadIdx.x;

  // ---- which pages: plain hybrid streams (with or without the 4-byte length / 1-byte width in front).  The prefix is
  // requested here and looked at behind the loads of the staging step, which do not wait for it: they cover what the
  // stream can be at most (the rest of the page, or RS_MAX_BYTES), and what lies behind its true end is cleared after.
  uint32_t pos = 0, w = d.bit_width, consumed_override = ~0u;
  const uint32_t n_values = d.num_values;
  bool mine = d.encoding == PQD_ENC_RLE && d.elem_size == 4 && n_values != 0 && n_values <= RS_MAX_VALUES;
  const PQD_GLOBAL uint8_t* gdata = (const PQD_GLOBAL uint8_t*)d.data;
  uint32_t prefix = 0;
  if (mine && (d.flags & PQD_F_WIDTH_PREFIX)) {
    mine = d.data_len != 0;
    if (mine) {
      prefix = gdata[0];
      pos = 1;
    }
  } else if (mine && (d.flags & PQD_F_LENGTH_PREFIX)) {
    mine = d.data_len >= 4;
    if (mine) {
      prefix = (uint32_t)gdata[0] | ((uint32_t)gdata[1] << 8) | ((uint32_t)gdata[2] << 16) | ((uint32_t)gdata[3] << 24);
      pos = 4;
    }
  }
  mine = mine && d.data_len > pos;
  if (!mine) {
    if (tid == 0) plans[page].flags = 0;
    return;
  }
  const uint32_t n_max = d.data_len - pos < RS_MAX_BYTES ? d.data_len - pos : RS_MAX_BYTES;
  const PQD_GLOBAL uint8_t* src = gdata + pos;
#ifdef PQD_PROF
  uint64_t stamp[10];
  int n_stamp = 0;
#define RS_STAMP() stamp[n_stamp++] = __builtin_readcyclecounter()
#else
#define RS_STAMP()
#endif
  RS_STAMP();

  // ---- 1. the stream -> LDS (zeros behind it).  Whole aligned dwords are read (none lies wholly outside the page) and
  // shifted into place: a piece of 16 stream bytes is five independent loads, whatever the page's address.
  constexpr int RS_PIECES = (RS_MAX_BYTES + 16 + RS_THREADS * 16 - 1) / (RS_THREADS * 16);  // 16-byte pieces per thread
  const uint32_t off = (uint32_t)((uintptr_t)src & 3u), sh = off * 8u;
  const PQD_GLOBAL uint32_t* wsrc = (const PQD_GLOBAL uint32_t*)((uintptr_t)src & ~(uintptr_t)3);
  uint32_t x[RS_PIECES][5];
  {
    const uint32_t words_max = (off + n_max + 3u) >> 2;  // dwords that hold a byte of what the stream can be at most
#pragma unroll
    for (int it = 0; it < RS_PIECES; ++it)
#pragma unroll
      for (uint32_t k = 0; k < 5u; ++k) {
        const uint32_t i = tid * 4u + (uint32_t)it * (RS_THREADS * 4u) + k;
        x[it][k] = i < words_max ? wsrc[i] : 0u;
      }
  }
  uint32_t end = d.data_len;
  if (d.flags & PQD_F_WIDTH_PREFIX) {
    w = prefix;
  } else if (d.flags & PQD_F_LENGTH_PREFIX) {
    mine = (int32_t)prefix >= 0 && prefix <= end - 4;
    end = 4 + prefix;
    consumed_override = end;
  }
  mine = mine && w >= 1 && w <= RS_MAX_WIDTH && end - pos <= RS_MAX_BYTES && end > pos;
  if (!mine) {
    if (tid == 0) plans[page].flags = 0;
    return;
  }
  const uint32_t N = end - pos;
  const uint32_t vb = (w + 7u) >> 3;  // bytes of an RLE run's value
  {
    const uint32_t full = (off + N) >> 2;  // dwords wholly inside [aligned start, end of stream)
    const uint32_t tail = (off + N) & 3u;  // stream bytes in dword `full`
#pragma unroll
    for (int it = 0; it < RS_PIECES; ++it) {
      const uint32_t i = tid * 4u + (uint32_t)it * (RS_THREADS * 4u);
      if (i * 4u >= N + 16u) continue;
#pragma unroll
      for (uint32_t k = 0; k < 5u; ++k) {
        if (i + k == full && tail)
          x[it][k] &= (1u << (8u * tail)) - 1u;
        else if (i + k >= full)
          x[it][k] = 0u;
      }
      const u32x4 v = {funnel_shr(x[it][0], x[it][1], sh), funnel_shr(x[it][1], x[it][2], sh), funnel_shr(x[it][2], x[it][3], sh),
                       funnel_shr(x[it][3], x[it][4], sh)};
      *reinterpret_cast<u32x4*>(pg + i * 4u) = v;
    }
  }
  if (tid < RS_BLOCKS) entry_of[tid] = 0xFFFFu;
  if (tid < RS_GROUPS) wbase[tid] = 0xFFFFu;  // (step 3's group entries; the word bases come later)
  if (tid == 0) {
    s_ok = 1u;
    s_consumed = 0u;
  }
  __syncthreads();
  const uint32_t* pgw = reinterpret_cast<const uint32_t*>(pg);
  const uint32_t block0 = tid * RS_BLOCK, block_end = block0 + RS_BLOCK;
  RS_STAMP();

  // ---- 2. candidates and block exits: a wave takes a 64-byte block at a time, one lane per byte.  Where a
  // chain that enters at a byte leaves the block: five rounds of "continue from where I land" between the
  // lanes (a run is at least two bytes, so 32 hops leave the block from anywhere).
  {
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_blocks = (N + RS_BLOCK) / RS_BLOCK;  // the block that holds position N is included
    constexpr int SIDE = 4;  // blocks worked on side by side (their chains of lane exchanges are independent)
    for (uint32_t b0 = wave * SIDE; b0 < n_blocks; b0 += SIDE * (RS_THREADS / 64)) {
      uint32_t to[SIDE];
#pragma unroll
      for (int q = 0; q < SIDE; ++q) {
        const uint32_t p = (b0 + q) * RS_BLOCK + lane;  // blocks past n_blocks: positions past N, complex, never stored
        const uint32_t pw = p < N + 4u ? p >> 2 : 0u;  // (positions past the stream read its start: they are complex anyway)
        const uint32_t lo = funnel_shr(pgw[pw], pgw[pw + 1], (p & 3u) * 8u);
        const SmallRun r = small_run(lo, w, vb);
        // where the run leads, counted from the block's first byte; RS_COMPLEX and RS_END are above every position
        to[q] = (r.complex || p + r.len > N) ? RS_COMPLEX : lane + r.len;
        if (p == N) to[q] = RS_END;
      }
#pragma unroll
      for (int round = 0; round < 5; ++round) {
        uint32_t further[SIDE];
#pragma unroll
        for (int q = 0; q < SIDE; ++q) further[q] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((to[q] & 63u) << 2), (int)to[q]);
#pragma unroll
        for (int q = 0; q < SIDE; ++q) to[q] = to[q] < RS_BLOCK ? further[q] : to[q];
      }
#pragma unroll
      for (int q = 0; q < SIDE; ++q) {
        const uint32_t p = (b0 + q) * RS_BLOCK + lane;
        const uint32_t abs_to = to[q] >= RS_END ? (to[q] == RS_END ? N : RS_COMPLEX) : (b0 + q) * RS_BLOCK + to[q];
        if (p <= N) tab[p] = (uint16_t)abs_to;
      }
    }
  }
  RS_STAMP();
  __syncthreads();
  RS_STAMP();
  // ---- 3. the chain.  One thread hopping from block to block is a chain of dependent LDS reads as long as the stream
  // has blocks, so: (a) for every group of eight blocks and every position of its first block, where a chain that
  // enters there leaves the group (four such chains per thread, eight hops at most, all threads at once); (b) one
  // thread hops from group to group (block by block only where a long run lands past a group's first block); (c) a
  // thread per group notes where the chain enters each of its blocks.
  uint16_t* gexit = reinterpret_cast<uint16_t*>(flags);  // RS_GROUPS x 64 exits; the bitmap is zeroed after this step
  uint16_t* gentry = wbase;
  static_assert(sizeof(flags) >= RS_GROUPS * RS_BLOCK * sizeof(uint16_t), "the group exits fit where the bitmap will be");
  {
    uint32_t p[4], lim[4];
#pragma unroll
    for (uint32_t c = 0; c < 4u; ++c) {
      const uint32_t idx = tid + c * RS_THREADS, g = idx / RS_BLOCK;
      p[c] = g * RS_GROUP + (idx % RS_BLOCK);
      lim[c] = (g + 1u) * RS_GROUP < N ? (g + 1u) * RS_GROUP : N;
    }
#pragma unroll
    for (int hop = 0; hop < 8; ++hop)
#pragma unroll
      for (uint32_t c = 0; c < 4u; ++c)
        if (p[c] < lim[c]) p[c] = tab[p[c]];  // (RS_COMPLEX is above every limit)
#pragma unroll
    for (uint32_t c = 0; c < 4u; ++c) gexit[tid + c * RS_THREADS] = (uint16_t)p[c];
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t p = 0;
    while (p < N) {
      if (p % RS_GROUP < RS_BLOCK) {
        gentry[p / RS_GROUP] = (uint16_t)p;
        p = gexit[(p / RS_GROUP) * RS_BLOCK + p % RS_BLOCK];
      } else {
        entry_of[p / RS_BLOCK] = (uint16_t)p;
        p = tab[p];
      }
    }
    if (p != N) s_ok = 0u;  // RS_COMPLEX met (a chain position beyond N cannot occur)
  }
  __syncthreads();
  if (!s_ok) {
    if (tid == 0) plans[page].flags = 0;
    return;
  }
  if (tid < RS_GROUPS) {
    uint32_t p = gentry[tid];
    const uint32_t lim = (tid + 1u) * RS_GROUP < N ? (tid + 1u) * RS_GROUP : N;
    while (p < lim) {  // (no entry: 0xFFFF)
      entry_of[p / RS_BLOCK] = (uint16_t)p;
      p = tab[p];
    }
  }
  for (uint32_t i = tid; i < (n_values >> 5) + 2u; i += RS_THREADS) flags[i] = 0u;
  __syncthreads();
  RS_STAMP();
  // ---- 4. the runs of every block, numbered, with their first output value
  uint32_t my_runs = 0, my_vals = 0;
  const uint32_t entry = tid < RS_BLOCKS ? entry_of[tid] : 0xFFFFu;
  if (entry != 0xFFFFu) {
    uint32_t p = entry;
    while (p < block_end && p < N) {
      const uint32_t lo = funnel_shr(pgw[p >> 2], pgw[(p >> 2) + 1], (p & 3u) * 8u);
      const SmallRun r = small_run(lo, w, vb);
      ++my_runs;
      my_vals += r.nvals < RS_MAX_VALUES ? r.nvals : RS_MAX_VALUES;
      p += r.len;
    }
  }
  uint32_t total_runs, total_vals;
  uint32_t run_i = rs_exclusive(my_runs, wave_totals, total_runs);
  uint32_t val_i = rs_exclusive(my_vals, wave_totals, total_vals);
  if (total_vals < n_values) {  // "Reading past RLE/BitPacking stream.": the other kernels report it
    if (tid == 0) plans[page].flags = 0;
    return;
  }
  uint32_t* runs = reinterpret_cast<uint32_t*>(tab);  // nobody reads the exits any more (two barriers above)
  if (entry != 0xFFFFu) {
    uint32_t p = entry;
    while (p < block_end && p < N) {
      const uint32_t lo = funnel_shr(pgw[p >> 2], pgw[(p >> 2) + 1], (p & 3u) * 8u);
      const SmallRun r = small_run(lo, w, vb);
      if (val_i < n_values) {
        // table entry: bit 31 set = bit-packed, then bits 0..30 = RS_BIAS + stream bit of the run's first
        // value - first value * w (value v of the run sits at entry + v * w - RS_BIAS); else the repeated value, its
        // vb bytes as they are (the reference does not mask them to the width)
        const uint32_t q = p + r.nb;
        const uint32_t rep = funnel_shr(pgw[q >> 2], pgw[(q >> 2) + 1], (q & 3u) * 8u) & ((1u << (8u * vb)) - 1u);
        runs[run_i] = r.packed ? (0x80000000u | (RS_BIAS + q * 8u - val_i * w)) : rep;
        atomicOr(&flags[val_i >> 5], 1u << (val_i & 31u));
        const uint32_t nv = r.nvals < RS_MAX_VALUES ? r.nvals : RS_MAX_VALUES;
        if (val_i + nv >= n_values) s_consumed = p + r.len;  // the run that holds the last value: read whole (:95-99)
      }
      ++run_i;
      val_i += r.nvals < RS_MAX_VALUES ? r.nvals : RS_MAX_VALUES;
      p += r.len;
    }
  }
  __syncthreads();
  RS_STAMP();
  // run starts before each 64-bit word of the bitmap
  {
    const uint32_t words = (n_values + 63u) >> 6;  // <= 512: two per thread
    uint32_t c[2], sum = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const uint32_t wd = tid * 2u + k;
      c[k] = wd < words ? (uint32_t)(__builtin_popcount(flags[2 * wd]) + __builtin_popcount(flags[2 * wd + 1])) : 0u;
      sum += c[k];
    }
    uint32_t total;
    uint32_t before = rs_exclusive(sum, wave_totals, total);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      wbase[tid * 2u + k] = (uint16_t)before;
      before += c[k];
    }
  }
  __syncthreads();
  RS_STAMP();
  // ---- 5. values, four per thread and step
  const uint32_t mask = (1u << w) - 1u;
  PQD_GLOBAL uint32_t* out = (PQD_GLOBAL uint32_t*)d.out;
#pragma unroll 2
  for (uint32_t v0 = tid * 4u; v0 < n_values; v0 += RS_THREADS * 4u) {
    const uint32_t wd = v0 >> 6, sh = v0 & 63u;  // the four values share a word of the bitmap
    const uint64_t bits = (uint64_t)flags[2 * wd] | ((uint64_t)flags[2 * wd + 1] << 32);
    uint32_t ri = wbase[wd] + (uint32_t)__builtin_popcountll(bits & ((2ull << sh) - 1ull)) - 1u;
    const uint32_t near = (uint32_t)(bits >> sh);  // bits 1..3: a run starts at value v0 + 1..3
    uint32_t val[4];
    uint32_t vw = v0 * w - RS_BIAS;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k) ri += (near >> k) & 1u;
      const uint32_t e = runs[ri];
      const uint32_t bit = (e & 0x7FFFFFFFu) + vw;
      const uint32_t i0 = (bit >> 5) & (RS_MAX_BYTES / 4 - 1u);  // (an RLE entry gives a meaningless but harmless index)
      const uint32_t x = funnel_shr(pgw[i0], pgw[i0 + 1], bit & 31u) & mask;
      val[k] = (e >> 31) ? x : e;
      vw += w;
    }
    if (v0 + 4u <= n_values) {
      const u32x4_a4 q = {val[0], val[1], val[2], val[3]};
      PQD_STREAM_STORE((PQD_GLOBAL u32x4_a4*)(out + v0), q);
    } else {
      for (uint32_t k = 0; v0 + k < n_values; ++k) out[v0 + k] = val[k];
    }
  }
  if (tid == 0) {
    pqd_page_result r = {PQD_OK, n_values, consumed_override != ~0u ? consumed_override : pos + s_consumed, 0u};
    results[page] = r;
    plans[page].flags = PLAN_F_SMALL_DONE;
  }
#ifdef PQD_PROF
  __syncthreads();
  RS_STAMP();
  if (tid == 0)  // development build: the page's first values are replaced by the cycles of each step
    for (int k = 0; k + 1 < n_stamp; ++k) out[k] = (uint32_t)(stamp[k + 1] - stamp[k]);
#endif
}

hipError_t launch_rle_small_pages(const pqd_page_desc* descs, pqd_page_result* results, void* plans, uint32_t n_pages,
                                  hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  hipLaunchKernelGGL(k_rle_small_pages, dim3(n_pages), dim3(RS_THREADS), 0, stream, descs, results,
                     reinterpret_cast<RlePagePlan*>(plans), n_pages);
  return hipGetLastError();
}

}  // namespace pqd
