// Expansion of planned RLE / bit-packing hybrid pages into int32 values.
//
// Does the work of RunLengthBitPackingHybridDecoder.readInt (parquet-column/.../values/rle/
// RunLengthBitPackingHybridDecoder.java:61-78) for every value of every page, but from the run plan
// that k_rle_plan_chunks left in device memory (pqd_rle_plan.hpp), so nothing here is serial: a
// chunk of the plan describes up to 32768 consecutive output values of one page by at most 231
// run segments.  Workgroups take chunks blockIdx.x, blockIdx.x + gridDim.x, ...; a lane owns
// units of four consecutive values: it finds the unit's segment (block table + a short walk),
// loads the unit's packed dwords with one or two loads, extracts the four values with funnel
// shifts / bit-field extracts and stores them with one 16-byte store.
// HBM traffic per value: w/8 bytes in, 4 bytes out; the plan adds 2 KiB per chunk.
#include "pqd_device.hpp"
#include "pqd_kernels.hpp"
#include "pqd_rle_plan.hpp"
#include "pqd_expand.hpp"

namespace pqd {

__device__ __forceinline__ uint32_t uni(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }

// One tile of a REGULAR chunk: every segment is a whole bit-packed run of `run_values` values and
// `run_stride` stream bytes, one behind the other (the plan kernel's bulk path marks such chunks).
// The run of a unit follows from its index, so there is no table, no search and no LDS traffic:
// a division by the (wave-uniform) run length, one load, the extracts, one store.
template <bool WIDE, int VPU>  // VPU: values per unit, 4, or 8 for widths up to 8 bits
__device__ __forceinline__ void expand_regular_tile(const ExHead& h, uint32_t start0, uint32_t src0, uint32_t run_values,
                                                    uint32_t run_stride, float rcp_values, uint32_t t0, uint32_t v1,
                                                    PQD_GLOBAL uint32_t* out, uint32_t tid) {
  static_assert(VPU == 4 || (VPU == 8 && !WIDE), "eight values per unit only where they fit 64 bits");
  constexpr int NW = WIDE ? 5 : 3;
  const uint32_t max_wi = h.last_word + 1 - NW;  // a regular run is at least 24 bytes: the page holds a unit load
  uint32_t wi[EX_UNITS], sh[EX_UNITS], delta[EX_UNITS];
#pragma unroll
  for (int k = 0; k < EX_UNITS; ++k) {
    const uint32_t g0 = t0 + VPU * ((uint32_t)k * EX_THREADS + tid);
    const uint32_t rel = (g0 < v1 ? g0 : t0) - start0;  // below 2^17: exact as a float
    uint32_t q = (uint32_t)((float)rel * rcp_values);
    q -= q * run_values > rel ? 1u : 0u;
    q += (q + 1u) * run_values <= rel ? 1u : 0u;
    const uint32_t byte0 = h.in_off + src0 + q * run_stride;
    const uint32_t bitrel = (byte0 & 3u) * 8 + (rel - q * run_values) * h.w;
    const uint32_t w_i = (byte0 >> 2) + (bitrel >> 5);
    delta[k] = w_i > max_wi ? w_i - max_wi : 0u;  // the page's last dwords: load what is there, shift below
    wi[k] = w_i - delta[k];
    sh[k] = bitrel & 31u;
  }
  uint32_t d[EX_UNITS][NW];
#pragma unroll
  for (int k = 0; k < EX_UNITS; ++k) {
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
  }
  const uint32_t mask = mask32(h.w);
#pragma unroll
  for (int k = 0; k < EX_UNITS; ++k) {
    const uint32_t g0 = t0 + VPU * ((uint32_t)k * EX_THREADS + tid);
    if (delta[k]) {  // rare
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        uint32_t x = 0;
#pragma unroll
        for (int i = j + 1; i < NW; ++i) x = (uint32_t)(i - j) == delta[k] ? d[k][i] : x;
        d[k][j] = x;
      }
    }
    uint32_t a[VPU];
    if (WIDE) {
      Window5 wnd = {d[k][0], d[k][1], d[k][2], d[k][WIDE ? 3 : 0], d[k][WIDE ? 4 : 0]};
      wnd.shr(sh[k]);
      a[0] = wnd.d0 & mask;
      wnd.drop(h.w);
      a[1] = wnd.d0 & mask;
      wnd.drop(h.w);
      a[2] = wnd.d0 & mask;
      wnd.drop(h.w);
      a[3] = wnd.d0 & mask;
    } else {
      const uint32_t lo = funnel_shr(d[k][0], d[k][1], sh[k]), hi = funnel_shr(d[k][1], d[k][2], sh[k]);
#pragma unroll
      for (int j = 0; j < VPU; ++j) a[j] = ex_unit_value(lo, hi, h.w, (uint32_t)j);
    }
    if (g0 < v1) {  // v1 of a regular chunk is a multiple of SEG_BLOCK: whole units only
#pragma unroll
      for (int j = 0; j < VPU; j += 4) {
        const u32x4_a4 r4 = {a[j], a[j + 1], a[j + 2], a[j + 3]};
        PQD_STREAM_STORE((PQD_GLOBAL u32x4_a4*)(out + g0 + j), r4);
      }
    }
  }
}

__global__ __launch_bounds__(EX_THREADS) void k_rle_expand(const RleSeg* __restrict__ pool, uint32_t pool_chunks,
                                                           const uint32_t* __restrict__ cursor, uint32_t n_pages) {
  __shared__ RleSeg cs_buf[SEG_CHUNK];
  __shared__ uint32_t slow_list[2 * EX_THREADS * EX_UNITS];  // {first value, segment} of the units left to the value-by-value pass
  __shared__ uint32_t slow_n[2];                              // their number, one counter per tile parity
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = uni(tid >> 6);
  // chunks to take: one per page (empty ones included) and the extra ones the plan kernel took from
  // the shared tail (it may have asked for more than the pool holds)
  uint32_t n_chunks = n_pages + uni(*cursor);
  if (n_chunks > pool_chunks) n_chunks = pool_chunks;
  const PQD_LDS RleSeg* cs = (const PQD_LDS RleSeg*)cs_buf;
  uint32_t parity = 0;
  if (tid < 2) slow_n[tid] = 0;  // (the first barrier of the chunk loop comes before any use)
  for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    __syncthreads();  // everybody is done with the previous chunk
    {
      const PQD_GLOBAL uint64_t* src = (const PQD_GLOBAL uint64_t*)(pool + (size_t)c * SEG_CHUNK);
      ((PQD_LDS uint64_t*)cs_buf)[tid] = src[tid];  // SEG_CHUNK == EX_THREADS slots of 8 bytes
    }
    __syncthreads();
    const PQD_LDS RleChunkHead* hd = (const PQD_LDS RleChunkHead*)(cs + SEG_HEAD_SLOT);
    const uint32_t dlo = uni((uint32_t)hd->data), dhi = uni((uint32_t)(hd->data >> 32));
    const uint64_t data = ((uint64_t)dhi << 32) | dlo;
    const uint64_t out_addr = ((uint64_t)uni((uint32_t)(hd->out >> 32)) << 32) | uni((uint32_t)hd->out);
    const uint32_t v0 = uni(hd->v0), v1 = uni(hd->v1);
    const uint32_t data_len = uni(hd->data_len), stream_end = uni(hd->stream_end);
    ExHead h;
    h.in_off = dlo & 3u;
    h.words = (const PQD_GLOBAL uint32_t*)(data - h.in_off);
    h.last_word = data_len ? (h.in_off + data_len - 1) >> 2 : 0u;
    const uint32_t end_bytes = h.in_off + stream_end;
    h.full_words = end_bytes >> 2;
    h.tail_mask = (end_bytes & 3u) ? ((1u << (8 * (end_bytes & 3u))) - 1u) : 0u;
    h.w = uni(hd->w);
    PQD_GLOBAL uint32_t* out = (PQD_GLOBAL uint32_t*)out_addr;
    const uint32_t run_values = uni(hd->reserved[0]);
    if (run_values) {  // a chunk of equal bit-packed runs
      const uint32_t run_stride = uni(hd->reserved[1]);
      const uint32_t start0 = uni(cs[0].start4) >> 2, src0 = uni(cs[0].src);
      const float rcp_values = 1.0f / (float)run_values;
      // (units of eight values for narrow widths were measured: slower, the stores coalesce worse)
      if (h.w <= 16) {
        for (uint32_t t0 = v0; t0 < v1; t0 += EX_TILE)
          expand_regular_tile<false, 4>(h, start0, src0, run_values, run_stride, rcp_values, t0, v1, out, tid);
      } else {
        for (uint32_t t0 = v0; t0 < v1; t0 += EX_TILE)
          expand_regular_tile<true, 4>(h, start0, src0, run_values, run_stride, rcp_values, t0, v1, out, tid);
      }
    } else if (h.w <= 16) {
      for (uint32_t t0 = v0; t0 < v1; t0 += EX_TILE, parity ^= 1u)
        expand_tile<false>(cs, h, v0, t0, v1, out, tid, wave, slow_list, slow_n, parity);
    } else {
      for (uint32_t t0 = v0; t0 < v1; t0 += EX_TILE, parity ^= 1u)
        expand_tile<true>(cs, h, v0, t0, v1, out, tid, wave, slow_list, slow_n, parity);
    }
  }
}

static int g_ex_cu_count = 0;

hipError_t launch_rle_expand(const void* pool, uint32_t pool_chunks, const uint32_t* cursor, uint32_t n_pages,
                             hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  if (g_ex_cu_count == 0) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return e;
    g_ex_cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  hipLaunchKernelGGL(k_rle_expand, dim3((uint32_t)g_ex_cu_count * 8), dim3(EX_THREADS), 0, stream,
                     reinterpret_cast<const RleSeg*>(pool), pool_chunks, cursor, n_pages);
  return hipGetLastError();
}

}  // namespace pqd
