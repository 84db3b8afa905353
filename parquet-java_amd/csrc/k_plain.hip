// PLAIN pages of fixed-width values and PLAIN booleans.
//
// Fixed width replaces PlainValuesReader.{Integer,Long,Float,Double}PlainValuesReader (parquet-column/
// .../values/plain/PlainValuesReader.java:38-145: initFromPage slices everything that is available, each
// read takes the next little-endian value, a read past the end is a ParquetDecodingException) and
// FixedLenByteArrayPlainValuesReader for elements of any fixed length: the decoded values ARE the page
// bytes, so this is a copy of min(num_values, available / elem_size) elements.
// Booleans replace BooleanPlainValuesReader (.../plain/BooleanPlainValuesReader.java:38-89): value i is
// bit (i & 7) of byte (i >> 3); the page holds min(ceil(valueCount / 8), available) bytes and a read
// beyond length * 8 bits fails.  Output: one byte (0 / 1) per value, or one int32 when elem_size is 4.
// HBM-bound: elem bytes in + elem bytes out per value (booleans: 1/8 byte in).
#include "pqd_device.hpp"
#include "pqd_kernels.hpp"

namespace pqd {

constexpr int PLAIN_THREADS = 256;

// (Pages run along grid.x, `per_page` workgroups each: grid.y and grid.z stop at 65 535 blocks, a batch does not.)
__global__ __launch_bounds__(PLAIN_THREADS) void k_plain_pages(const pqd_page_desc* __restrict__ descs,
                                                               pqd_page_result* __restrict__ results, uint32_t n_pages,
                                                               uint32_t per_page) {
  const uint32_t page = blockIdx.x / per_page, part = blockIdx.x % per_page;
  if (page >= n_pages) return;
  const pqd_page_desc d = descs[page];
  const uint32_t K = d.elem_size;
  const bool lead = part == 0 && threadIdx.x == 0;
  if (d.encoding != PQD_ENC_PLAIN || K == 0 || K > 16) {
    if (lead) {
      pqd_page_result r = {PQD_BAD_ARG, 0u, 0u, 0u};
      results[page] = r;
    }
    return;
  }
  const uint32_t have = d.data_len / K;
  const uint32_t n = d.num_values < have ? d.num_values : have;
  const uint64_t bytes = (uint64_t)n * K;
  const uint8_t* in = d.data;
  uint8_t* out = reinterpret_cast<uint8_t*>(d.out);
  // 16-byte pieces from any source address to any destination address (4-byte aligned accesses are
  // not required on this device; the pieces of a lane's neighbours are adjacent)
  const uint64_t pieces = bytes >> 4;
  const uint64_t stride = (uint64_t)per_page * PLAIN_THREADS;
  for (uint64_t i = (uint64_t)part * PLAIN_THREADS + threadIdx.x; i < pieces; i += stride) {
    u32x4 v;
    __builtin_memcpy(&v, in + i * 16, 16);
    __builtin_memcpy(out + i * 16, &v, 16);
  }
  if (part == 0) {
    for (uint64_t b = (pieces << 4) + threadIdx.x; b < bytes; b += PLAIN_THREADS) out[b] = in[b];
  }
  if (lead) {
    // the reader took everything available (initFromPage :52-60); the read after the last whole
    // value fails ("could not read int")
    pqd_page_result r = {n < d.num_values ? PQD_DECODE : PQD_OK, n, d.data_len, 0u};
    results[page] = r;
  }
}

template <int ELEM>
__global__ __launch_bounds__(PLAIN_THREADS) void k_plain_boolean_pages(const pqd_page_desc* __restrict__ descs,
                                                                       pqd_page_result* __restrict__ results,
                                                                       uint32_t n_pages, uint32_t per_page) {
  const uint32_t page = blockIdx.x / per_page, part = blockIdx.x % per_page;
  if (page >= n_pages) return;
  const pqd_page_desc d = descs[page];
  const bool lead = part == 0 && threadIdx.x == 0;
  if (d.encoding != PQD_ENC_PLAIN_BOOLEAN || d.elem_size != ELEM) {
    if (lead) {
      pqd_page_result r = {PQD_BAD_ARG, 0u, 0u, 0u};
      results[page] = r;
    }
    return;
  }
  uint64_t length = ((uint64_t)d.num_values + 7) >> 3;  // BytesUtils.paddedByteCountFromBits(valueCount)
  if (length > d.data_len) length = d.data_len;          // Math.min(length, stream.available())
  const uint64_t bit_count = length * 8;
  const uint32_t n = d.num_values < bit_count ? d.num_values : (uint32_t)bit_count;
  const uint8_t* in = d.data;
  // one input byte = eight values per lane and step
  const uint32_t n_bytes = (n + 7) >> 3;
  const uint32_t stride = per_page * PLAIN_THREADS;
  for (uint32_t b = part * PLAIN_THREADS + threadIdx.x; b < n_bytes; b += stride) {
    const uint32_t x = in[b];
    const uint32_t first = b * 8;
    if (ELEM == 1) {
      // bit j -> byte j: spread the low and the high nibble over a dword each
      const uint32_t lo = ((x & 1u)) | ((x & 2u) << 7) | ((x & 4u) << 14) | ((x & 8u) << 21);
      const uint32_t hi = ((x >> 4) & 1u) | (((x >> 4) & 2u) << 7) | (((x >> 4) & 4u) << 14) | (((x >> 4) & 8u) << 21);
      uint8_t* o = reinterpret_cast<uint8_t*>(d.out) + first;
      if (first + 8 <= n) {
        const u32x2 v = {lo, hi};
        __builtin_memcpy(o, &v, 8);
      } else {
        for (uint32_t j = 0; first + j < n; ++j) o[j] = (uint8_t)((x >> j) & 1u);
      }
    } else {
      uint32_t* o = reinterpret_cast<uint32_t*>(d.out) + first;
      if (first + 8 <= n) {
        const u32x4_a4 v0 = {x & 1u, (x >> 1) & 1u, (x >> 2) & 1u, (x >> 3) & 1u};
        const u32x4_a4 v1 = {(x >> 4) & 1u, (x >> 5) & 1u, (x >> 6) & 1u, (x >> 7) & 1u};
        *reinterpret_cast<u32x4_a4*>(o) = v0;
        *reinterpret_cast<u32x4_a4*>(o + 4) = v1;
      } else {
        for (uint32_t j = 0; first + j < n; ++j) o[j] = (x >> j) & 1u;
      }
    }
  }
  if (lead) {
    // "attempt to read beyond end of boolean page" at the first value past length * 8 bits (:69-73)
    pqd_page_result r = {n < d.num_values ? PQD_DECODE : PQD_OK, n, (uint32_t)length, 0u};
    results[page] = r;
  }
}

static uint32_t workgroups_per_page(uint32_t n_pages) {
  // few pages: several workgroups share a page (grid-stride over its pieces)
  uint32_t per_page = n_pages >= 2048 ? 1u : (4096u + n_pages - 1) / n_pages;
  return per_page > 64 ? 64u : per_page;
}

hipError_t launch_plain_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages, hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  const uint32_t per_page = workgroups_per_page(n_pages);
  hipLaunchKernelGGL(k_plain_pages, dim3(per_page * n_pages), dim3(PLAIN_THREADS), 0, stream, descs, results, n_pages, per_page);
  return hipGetLastError();
}

hipError_t launch_plain_boolean_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages, int elem,
                                      hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  const uint32_t per_page = workgroups_per_page(n_pages);
  const dim3 grid(per_page * n_pages), block(PLAIN_THREADS);
  if (elem == 1)
    hipLaunchKernelGGL(k_plain_boolean_pages<1>, grid, block, 0, stream, descs, results, n_pages, per_page);
  else
    hipLaunchKernelGGL(k_plain_boolean_pages<4>, grid, block, 0, stream, descs, results, n_pages, per_page);
  return hipGetLastError();
}

}  // namespace pqd
