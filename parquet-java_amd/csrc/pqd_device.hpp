// Device-side helpers shared by the page-decode kernels (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pqd.h"

// Store of decoded output: written once, never read again by the kernels.  The nontemporal hint
// keeps those lines from settling in L2, where they would push out what the kernels do read again
// (dictionary slices, plan chunks) and fill the write-back capacity in bursts.
#define PQD_STREAM_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))

// Pointers that come out of descriptors in memory are generic to the compiler, and accesses through
// generic pointers are FLAT instructions: those count as LDS operations as well, so a kernel that waits
// for an LDS read then also waits for its outstanding stores.  Kernels that mix LDS and memory traffic
// cast such pointers to the global address space once.
#define PQD_GLOBAL __attribute__((address_space(1)))
#define PQD_LDS __attribute__((address_space(3)))

namespace pqd {

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ({hi,lo} >> sh) & 0xffffffff for sh in 0..31 (v_alignbit_b32; sh == 0 returns lo).
__device__ __forceinline__ uint32_t funnel_shr(uint32_t lo, uint32_t hi, uint32_t sh) {
  return __builtin_amdgcn_alignbit(hi, lo, sh);
}

// A 160-bit little-endian window held in five dwords; holds four fields of up to 32 bits (or two
// of up to 64) at any starting bit 0..31.
struct Window5 {
  uint32_t d0, d1, d2, d3, d4;
  __device__ __forceinline__ void shr(uint32_t s) {  // s in 0..31
    d0 = funnel_shr(d0, d1, s);
    d1 = funnel_shr(d1, d2, s);
    d2 = funnel_shr(d2, d3, s);
    d3 = funnel_shr(d3, d4, s);
    d4 = d4 >> s;
  }
  __device__ __forceinline__ void shr32() {
    d0 = d1;
    d1 = d2;
    d2 = d3;
    d3 = d4;
    d4 = 0;
  }
  // drop `w` bits, w in 0..64
  __device__ __forceinline__ void drop(uint32_t w) {
    if (w >= 32) {
      shr32();
      w -= 32;
    }
    if (w >= 32) {
      shr32();
      w -= 32;
    }
    shr(w);
  }
};

__device__ __forceinline__ uint32_t mask32(uint32_t w) {  // w in 0..32
  return w >= 32 ? 0xFFFFFFFFu : ((1u << w) - 1u);
}
__device__ __forceinline__ uint64_t mask64(uint32_t w) {  // w in 0..64
  return w >= 64 ? ~0ull : ((1ull << w) - 1ull);
}

// Stages `pieces` 16-byte pieces of LDS from global memory starting at the 16-byte-aligned address
// `gbase`; bytes at addresses >= `gend` are written as zero (so a truncated final run decodes
// zeros, and no load touches a 16-byte block that lies wholly past the end of the page).
// Four independent loads are issued per lane before the first is consumed.
__device__ __forceinline__ u32x4 keep_leading_bytes(u32x4 v, long valid) {
  if (valid >= 16) return v;
  if (valid <= 0) return u32x4{0u, 0u, 0u, 0u};
  const uint32_t full = (uint32_t)valid >> 2, rem = (uint32_t)valid & 3u;
  const uint32_t part = rem ? ((1u << (8 * rem)) - 1u) : 0u;
  v.x = full > 0 ? v.x : v.x & part;
  v.y = full > 1 ? v.y : (full == 1 ? v.y & part : 0u);
  v.z = full > 2 ? v.z : (full == 2 ? v.z & part : 0u);
  v.w = full > 3 ? v.w : (full == 3 ? v.w & part : 0u);
  return v;
}

template <int THREADS>
__device__ __forceinline__ void stage_window(u32x4* lds, const uint8_t* gbase, const uint8_t* gend,
                                             int pieces) {
  for (int i0 = threadIdx.x; i0 < pieces; i0 += THREADS * 4) {
    u32x4 v[4];
    long valid[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + j * THREADS;
      const uint8_t* p = gbase + (size_t)i * 16;
      valid[j] = i < pieces ? (long)(gend - p) : 0;
      v[j] = u32x4{0u, 0u, 0u, 0u};
      if (valid[j] > 0) v[j] = *(const PQD_GLOBAL u32x4*)(p);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + j * THREADS;
      if (i < pieces) lds[i] = keep_leading_bytes(v[j], valid[j]);
    }
  }
}

// Inclusive prefix sum over the 64 lanes of a wave with DPP moves (no LDS traffic): four shifts
// inside the rows of 16 lanes, then the row totals are passed on with the two row broadcasts.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_scan_inclusive(uint32_t v) {
  v += dpp_or_zero<0x111, 0xF>(v);  // row_shr:1
  v += dpp_or_zero<0x112, 0xF>(v);  // row_shr:2
  v += dpp_or_zero<0x114, 0xF>(v);  // row_shr:4
  v += dpp_or_zero<0x118, 0xF>(v);  // row_shr:8
  v += dpp_or_zero<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_or_zero<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ uint64_t wave_scan_inclusive(uint64_t v) {
#define PQD_SCAN_STEP64(CTRL, MASK)                                                                     \
  v += ((uint64_t)dpp_or_zero<CTRL, MASK>((uint32_t)(v >> 32)) << 32) | dpp_or_zero<CTRL, MASK>((uint32_t)v);
  PQD_SCAN_STEP64(0x111, 0xF)
  PQD_SCAN_STEP64(0x112, 0xF)
  PQD_SCAN_STEP64(0x114, 0xF)
  PQD_SCAN_STEP64(0x118, 0xF)
  PQD_SCAN_STEP64(0x142, 0xA)
  PQD_SCAN_STEP64(0x143, 0xC)
#undef PQD_SCAN_STEP64
  return v;
}

__device__ __forceinline__ uint32_t lds_byte(const uint32_t* lds_words, uint32_t off) {
  return (lds_words[off >> 2] >> ((off & 3u) * 8u)) & 0xFFu;
}

// Reads the five dwords that cover bit position `bitpos` onwards and aligns them to bit 0.
__device__ __forceinline__ Window5 load_window(const uint32_t* words, uint32_t bitpos) {
  uint32_t i = bitpos >> 5;
  Window5 wnd;
  wnd.d0 = words[i];
  wnd.d1 = words[i + 1];
  wnd.d2 = words[i + 2];
  wnd.d3 = words[i + 3];
  wnd.d4 = words[i + 4];
  wnd.shr(bitpos & 31u);
  return wnd;
}

}  // namespace pqd
