// Bulk bit-unpack of one bit-packed run (LSB-first), widths 1..32 -> int32 and 1..64 -> int64.
//
// Replaces the loop over BytePacker.unpack8Values / unpackValuesUsingVector in
// ParquetReadRouter.readBatchUsing512Vector (parquet-plugins/parquet-encoding-vector/.../
// ParquetReadRouter.java:69-104) and BytePackerForLong.unpack8Values over a run.  Value i is bits
// [i*w, (i+1)*w) of the byte stream read as one little-endian integer, which is what the generated
// ByteBitPackingLE classes compute (parquet-generator/.../ByteBasedBitPackingGenerator.java:136-182,258-308).
//
// HBM-bound: w/8 bytes in, 4 (or 8) bytes out per value.  Each lane owns four (two for int64)
// consecutive values so a wave stores 1 KiB contiguous per instruction; the packed dwords covering
// those values (at most five) are loaded with plain coalesced dword loads and consumed in registers
// with funnel shifts.
#include "pqd_device.hpp"
#include "pqd_kernels.hpp"

namespace pqd {

__global__ __launch_bounds__(256) void k_unpack_run_i32(const uint8_t* __restrict__ in, uint64_t in_bytes,
                                                        int32_t* __restrict__ out, uint64_t n_values,
                                                        uint32_t w) {
  const uint32_t in_off = (uint32_t)((uintptr_t)in & 3u);
  const uint32_t* words = reinterpret_cast<const uint32_t*>(in - in_off);
  const uint64_t n_words = (in_off + in_bytes + 3) >> 2;  // dwords that hold at least one real byte
  const uint32_t mask = mask32(w);
  const uint64_t n_units = n_values >> 2;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_units; u += stride) {
    const uint64_t bit = (uint64_t)in_off * 8 + u * 4 * w;
    const uint64_t wi = bit >> 5;
    const uint32_t sh = (uint32_t)bit & 31u;
    const uint32_t span = sh + 4 * w;  // bits needed from words[wi]
    Window5 wnd;
    wnd.d0 = words[wi];
    wnd.d1 = (span > 32 && wi + 1 < n_words) ? words[wi + 1] : 0u;
    wnd.d2 = (span > 64 && wi + 2 < n_words) ? words[wi + 2] : 0u;
    wnd.d3 = (span > 96 && wi + 3 < n_words) ? words[wi + 3] : 0u;
    wnd.d4 = (span > 128 && wi + 4 < n_words) ? words[wi + 4] : 0u;
    wnd.shr(sh);
    u32x4_a4 r;
    r.x = wnd.d0 & mask;
    wnd.drop(w);
    r.y = wnd.d0 & mask;
    wnd.drop(w);
    r.z = wnd.d0 & mask;
    wnd.drop(w);
    r.w = wnd.d0 & mask;
    PQD_STREAM_STORE(reinterpret_cast<u32x4_a4*>(out + u * 4), r);
  }
}

__global__ __launch_bounds__(256) void k_unpack_run_i64(const uint8_t* __restrict__ in, uint64_t in_bytes,
                                                        int64_t* __restrict__ out, uint64_t n_values,
                                                        uint32_t w) {
  const uint32_t in_off = (uint32_t)((uintptr_t)in & 3u);
  const uint32_t* words = reinterpret_cast<const uint32_t*>(in - in_off);
  const uint64_t n_words = (in_off + in_bytes + 3) >> 2;
  const uint64_t mask = mask64(w);
  const uint64_t n_units = n_values >> 1;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_units; u += stride) {
    const uint64_t bit = (uint64_t)in_off * 8 + u * 2 * w;
    const uint64_t wi = bit >> 5;
    const uint32_t sh = (uint32_t)bit & 31u;
    const uint32_t span = sh + 2 * w;
    Window5 wnd;
    wnd.d0 = words[wi];
    wnd.d1 = (span > 32 && wi + 1 < n_words) ? words[wi + 1] : 0u;
    wnd.d2 = (span > 64 && wi + 2 < n_words) ? words[wi + 2] : 0u;
    wnd.d3 = (span > 96 && wi + 3 < n_words) ? words[wi + 3] : 0u;
    wnd.d4 = (span > 128 && wi + 4 < n_words) ? words[wi + 4] : 0u;
    wnd.shr(sh);
    uint64_t a = (((uint64_t)wnd.d1 << 32) | wnd.d0) & mask;
    wnd.drop(w);
    uint64_t b = (((uint64_t)wnd.d1 << 32) | wnd.d0) & mask;
    u32x4_a4 r;
    r.x = (uint32_t)a;
    r.y = (uint32_t)(a >> 32);
    r.z = (uint32_t)b;
    r.w = (uint32_t)(b >> 32);
    PQD_STREAM_STORE(reinterpret_cast<u32x4_a4*>(out + u * 2), r);
  }
}


static uint32_t unpack_grid(uint64_t n_units) {
  uint64_t blocks = (n_units + 255) / 256;
  const uint64_t cap = 256 * 16;  // 16 workgroups per CU, grid-stride beyond that
  return (uint32_t)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

hipError_t launch_unpack_run_i32(const uint8_t* in, uint64_t in_bytes, int32_t* out, uint64_t n_values,
                                 uint32_t w, hipStream_t stream) {
  if (n_values == 0) return hipSuccess;
  hipLaunchKernelGGL(k_unpack_run_i32, dim3(unpack_grid(n_values >> 2)), dim3(256), 0, stream, in, in_bytes, out,
                     n_values, w);
  return hipGetLastError();
}

hipError_t launch_unpack_run_i64(const uint8_t* in, uint64_t in_bytes, int64_t* out, uint64_t n_values,
                                 uint32_t w, hipStream_t stream) {
  if (n_values == 0) return hipSuccess;
  hipLaunchKernelGGL(k_unpack_run_i64, dim3(unpack_grid(n_values >> 1)), dim3(256), 0, stream, in, in_bytes, out,
                     n_values, w);
  return hipGetLastError();
}

}  // namespace pqd
