// BYTE_STREAM_SPLIT page decode and legacy BIT_PACKED level decode.
//
// BYTE_STREAM_SPLIT replaces ByteStreamSplitValuesReader.initFromPage/decodeData
// (parquet-column/.../values/bytestreamsplit/ByteStreamSplitValuesReader.java:53-218):
// n = available / K values, out[i*K + s] = in[s*n + i].  A pure byte transpose; floating-point
// payloads are never touched arithmetically, so NaN payloads and signed zeros survive.
// HBM-bound: K bytes in + K bytes out per value.  Each lane owns four consecutive values: it
// loads one (possibly unaligned) dword from each of the K streams, transposes the 4 x K bytes in
// registers with byte permutes and stores 4*K contiguous bytes.
//
// BIT_PACKED replaces ByteBitPackingValuesReader with Packer.BIG_ENDIAN
// (parquet-column/.../values/bitpacking/ByteBitPackingValuesReader.java:41-88): MSB-first fields,
// ceil(n*w/8) bytes clipped to the stream, missing bytes read as zero.
#include "pqd_device.hpp"
#include "pqd_kernels.hpp"

namespace pqd {

constexpr int BSS_THREADS = 256;

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}

// 4x4 byte transpose: in x[s] = bytes of stream s for values 0..3; out y[j] = bytes 0..3 of value j
__device__ __forceinline__ void transpose4x4(const uint32_t x[4], uint32_t y[4]) {
  // v_perm_b32(S0, S1, sel): selector 0-3 pick bytes of S1, 4-7 bytes of S0
  const uint32_t a01 = __builtin_amdgcn_perm(x[1], x[0], 0x05010400u);  // x0.b0 x1.b0 x0.b1 x1.b1
  const uint32_t b01 = __builtin_amdgcn_perm(x[1], x[0], 0x07030602u);  // x0.b2 x1.b2 x0.b3 x1.b3
  const uint32_t a23 = __builtin_amdgcn_perm(x[3], x[2], 0x05010400u);
  const uint32_t b23 = __builtin_amdgcn_perm(x[3], x[2], 0x07030602u);
  y[0] = __builtin_amdgcn_perm(a23, a01, 0x05040100u);  // a01.b0 a01.b1 a23.b0 a23.b1
  y[1] = __builtin_amdgcn_perm(a23, a01, 0x07060302u);
  y[2] = __builtin_amdgcn_perm(b23, b01, 0x05040100u);
  y[3] = __builtin_amdgcn_perm(b23, b01, 0x07060302u);
}

__global__ __launch_bounds__(BSS_THREADS) void k_bss_pages(const pqd_page_desc* __restrict__ descs,
                                                           pqd_page_result* __restrict__ results,
                                                           uint32_t n_pages) {
  const uint32_t page = blockIdx.y;
  if (page >= n_pages) return;
  const pqd_page_desc d = descs[page];
  const uint32_t K = d.elem_size;
  const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
  int32_t status = PQD_OK;
  if (d.encoding != PQD_ENC_BYTE_STREAM_SPLIT || K == 0 || K > 16)
    status = PQD_BAD_ARG;
  else if (d.data_len % K != 0)
    status = PQD_DECODE;  // :195-200
  const uint32_t n = status == PQD_OK ? d.data_len / K : 0;
  if (status == PQD_OK && d.num_values < n) status = PQD_DECODE;  // :204-209
  if (status != PQD_OK) {
    if (lead) {
      pqd_page_result r = {status, 0u, 0u, 0u};
      results[page] = r;
    }
    return;
  }
  const uint8_t* in = d.data;
  uint8_t* out = reinterpret_cast<uint8_t*>(d.out);
  const uint32_t n_units = n >> 2;
  const uint32_t stride = gridDim.x * BSS_THREADS;
  if (K == 4) {
    for (uint32_t u = blockIdx.x * BSS_THREADS + threadIdx.x; u < n_units; u += stride) {
      uint32_t x[4], y[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) x[s] = load_u32_unaligned(in + (size_t)s * n + (size_t)u * 4);
      transpose4x4(x, y);
      u32x4_a4 r = {y[0], y[1], y[2], y[3]};
      PQD_STREAM_STORE(reinterpret_cast<u32x4_a4*>(out + (size_t)u * 16), r);
    }
  } else if (K == 8) {
    for (uint32_t u = blockIdx.x * BSS_THREADS + threadIdx.x; u < n_units; u += stride) {
      uint32_t x[8], lo[4], hi[4];
#pragma unroll
      for (int s = 0; s < 8; ++s) x[s] = load_u32_unaligned(in + (size_t)s * n + (size_t)u * 4);
      transpose4x4(x, lo);
      transpose4x4(x + 4, hi);
      u32x4_a4 r0 = {lo[0], hi[0], lo[1], hi[1]};
      u32x4_a4 r1 = {lo[2], hi[2], lo[3], hi[3]};
      *reinterpret_cast<u32x4_a4*>(out + (size_t)u * 32) = r0;
      *reinterpret_cast<u32x4_a4*>(out + (size_t)u * 32 + 16) = r1;
    }
  } else {
    // FIXED_LEN_BYTE_ARRAY(K): generic per-byte transpose (:176-183)
    for (uint32_t u = blockIdx.x * BSS_THREADS + threadIdx.x; u < n_units; u += stride)
      for (uint32_t j = 0; j < 4; ++j)
        for (uint32_t s = 0; s < K; ++s)
          out[((size_t)u * 4 + j) * K + s] = in[(size_t)s * n + (size_t)u * 4 + j];
  }
  // tail (n % 4 values)
  if (blockIdx.x == 0) {
    const uint32_t tail0 = n_units << 2;
    for (uint32_t i = tail0 + threadIdx.x / 16; i < n; i += BSS_THREADS / 16) {
      const uint32_t s = threadIdx.x & 15u;
      if (s < K) out[(size_t)i * K + s] = in[(size_t)s * n + i];
    }
  }
  if (lead) {
    pqd_page_result r = {PQD_OK, n, d.data_len, 0u};
    results[page] = r;
  }
}

// Legacy BIT_PACKED levels, MSB-first.  One workgroup per page, one value per lane per step.
__global__ __launch_bounds__(256) void k_bitpacked_pages(const pqd_page_desc* __restrict__ descs,
                                                         pqd_page_result* __restrict__ results,
                                                         uint32_t n_pages) {
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_page_desc d = descs[page];
  const uint32_t w = d.bit_width;
  if (d.encoding != PQD_ENC_BIT_PACKED || d.elem_size != 4 || w > 32) {
    if (threadIdx.x == 0) {
      pqd_page_result r = {PQD_BAD_ARG, 0u, 0u, 0u};
      results[page] = r;
    }
    return;
  }
  const uint32_t n = d.num_values;
  uint64_t length = ((uint64_t)n * w + 7) >> 3;  // :78-79
  if (length > d.data_len) length = d.data_len;  // :84
  PQD_GLOBAL uint32_t* out = (PQD_GLOBAL uint32_t*)d.out;
  const PQD_GLOBAL uint8_t* data = (const PQD_GLOBAL uint8_t*)d.data;
  const uint32_t mask = mask32(w);
  // one value from the five bytes around it (a group whose bytes run past `length` is zero-filled, :48-57)
  auto one = [&](uint32_t i) {
    const uint64_t bit = (uint64_t)i * w;
    const uint64_t byte = bit >> 3;
    uint64_t acc = 0;  // big-endian 40-bit window
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const uint64_t b = byte + k;
      acc = (acc << 8) | (b < length ? (uint64_t)data[b] : 0ull);
    }
    const uint32_t sh = 40u - ((uint32_t)bit & 7u) - w;
    return (uint32_t)(acc >> sh) & mask;
  };
  // four values per lane and step; widths up to 8 (levels): the four lie within eight bytes read at once
  for (uint32_t i = threadIdx.x * 4u; i < n; i += 256u * 4u) {
    const uint64_t bit = (uint64_t)i * w;
    const uint64_t byte = bit >> 3;
    if (w <= 8u && i + 4u <= n && byte + 8u <= length) {
      uint64_t raw;
      __builtin_memcpy(&raw, data + byte, 8);
      const uint64_t be = __builtin_bswap64(raw);
      const uint32_t sh = 64u - ((uint32_t)bit & 7u) - w;
      const u32x4_a4 q = {(uint32_t)(be >> sh) & mask, (uint32_t)(be >> (sh - w)) & mask, (uint32_t)(be >> (sh - 2u * w)) & mask,
                          (uint32_t)(be >> (sh - 3u * w)) & mask};
      *(PQD_GLOBAL u32x4_a4*)(out + i) = q;
    } else {
      for (uint32_t k = 0; k < 4u && i + k < n; ++k) out[i + k] = one(i + k);
    }
  }
  if (threadIdx.x == 0) {
    pqd_page_result r = {PQD_OK, n, (uint32_t)length, 0u};
    results[page] = r;
  }
}


hipError_t launch_bss_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages,
                            hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  // few pages: several workgroups share a page (grid-stride over its 4-value units)
  uint32_t per_page = n_pages >= 2048 ? 1u : (4096u + n_pages - 1) / n_pages;
  if (per_page > 64) per_page = 64;
  hipLaunchKernelGGL(k_bss_pages, dim3(per_page, n_pages), dim3(BSS_THREADS), 0, stream, descs, results, n_pages);
  return hipGetLastError();
}

hipError_t launch_bitpacked_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages,
                                  hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  hipLaunchKernelGGL(k_bitpacked_pages, dim3(n_pages), dim3(256), 0, stream, descs, results, n_pages);
  return hipGetLastError();
}

}  // namespace pqd
