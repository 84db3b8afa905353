// What the one-wave-per-page kernels share (ZSTD, GZIP, the suffix stream of DELTA_BYTE_ARRAY): the input staged in LDS
// chunk by chunk (InStage) and an output ring with its copy primitives (OutRing).
#pragma once

#include "pqd_device.hpp"

namespace pqd {

constexpr uint32_t LZ_CHUNK = 2048;
constexpr uint32_t LZ_STAGE = 2 * LZ_CHUNK;

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// An order for the LDS accesses of ONE wave's lanes (the kernels that use this header run a page's serial parts in one
// wave; where a second wave of the workgroup does other work, a workgroup barrier would be wrong).  LDS instructions of a
// wave complete in order; the fences keep the compiler from moving accesses across.
__device__ __forceinline__ void lz_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (Memory is addressed through global-space pointers: see PQD_GLOBAL.)
struct InStage {
  const PQD_GLOBAL uint8_t* base;  // the source, rounded down to 16 bytes; positions below count from here
  uint32_t end;         // position behind the last source byte
  uint32_t cur;         // chunks cur and cur + 1 are in LDS, chunk cur + 2 in pre0 / pre1
  uint32_t lane;
  uint8_t* lds;
  u32x4 pre0, pre1;

  __device__ __forceinline__ void load(uint32_t c, u32x4& r0, u32x4& r1) const {
    const uint32_t p = c * LZ_CHUNK + lane * 32u;
    r0 = u32x4{0u, 0u, 0u, 0u};
    r1 = r0;
    if (p < end) r0 = *reinterpret_cast<const PQD_GLOBAL u32x4*>(base + p);
    if (p + 16u < end) r1 = *reinterpret_cast<const PQD_GLOBAL u32x4*>(base + p + 16u);
  }
  __device__ __forceinline__ void put(uint32_t c, u32x4 r0, u32x4 r1) {
    u32x4* d = reinterpret_cast<u32x4*>(lds + (c & 1u) * LZ_CHUNK + lane * 32u);
    d[0] = r0;
    d[1] = r1;
  }
  __device__ __forceinline__ void start(uint32_t c) {
    u32x4 a0, a1, b0, b1;
    load(c, a0, a1);
    load(c + 1, b0, b1);
    lz_sync();
    put(c, a0, a1);
    put(c + 1, b0, b1);
    load(c + 2, pre0, pre1);
    cur = c;
    lz_sync();
  }
  // makes the chunk of `pos` and the one behind it readable
  __device__ __forceinline__ void reach(uint32_t pos) {
    const uint32_t c = pos / LZ_CHUNK;
    if (c == cur) return;
    if (c == cur + 1) {
      lz_sync();
      put(cur + 2, pre0, pre1);
      cur = c;
      load(cur + 2, pre0, pre1);
      lz_sync();
    } else {
      start(c);
    }
  }
  __device__ __forceinline__ uint32_t byte(uint32_t pos) const { return lds[pos & (LZ_STAGE - 1u)]; }
  __device__ __forceinline__ uint32_t ubyte(uint32_t pos) const { return uni(byte(pos)); }

  // The element headers are read from a 16-byte window held in scalar registers: win0 = the eight
  // bytes at the 8-aligned position win_at, win1 the eight behind them.  The window moves on in steps
  // of eight bytes; the half that comes in is requested from LDS when the step is taken and is not
  // needed before the headers have moved past the half in front of it, so its latency is hidden.
  uint64_t win0, win1;
  uint32_t win_at;
  __device__ __forceinline__ uint64_t load8(uint32_t pos8) const {  // pos8: a multiple of 8 inside the staged chunks
    const u32x2 v = *reinterpret_cast<const u32x2*>(lds + (pos8 & (LZ_STAGE - 1u)));
    return ((uint64_t)uni(v.y) << 32) | uni(v.x);
  }
  // the eight bytes at `pos` (reach(pos) has been called; pos + 8 may lie one chunk further)
  __device__ __forceinline__ uint64_t peek8(uint32_t pos) {
    if (pos - win_at >= 16u) {  // a jump (long literal, first use): both halves anew
      win_at = pos & ~7u;
      win0 = load8(win_at);
      win1 = load8(win_at + 8u);
    } else if (pos - win_at >= 8u) {
      win_at += 8u;
      win0 = win1;
      win1 = load8(win_at + 8u);
    }
    const uint32_t sh = (pos - win_at) * 8u;
    return sh ? (win0 >> sh) | (win1 << (64u - sh)) : win0;
  }
};

struct OutRing {
  uint8_t* ring;
  PQD_GLOBAL uint8_t* dst;
  uint32_t lane;
  uint32_t ring_mask;  // ring bytes - 1

  // `len` literal bytes from the staged input (they lie within the staged chunks) or, long ones, from memory
  __device__ __forceinline__ void literal(InStage& in, uint32_t from, uint32_t op, uint32_t len,
                                          uint32_t staged_limit = LZ_CHUNK - 16u) {
    if (len <= staged_limit) {
      for (uint32_t i = lane; i < len; i += 64u) {
        const uint8_t v = (uint8_t)in.byte(from + i);
        ring[(op + i) & ring_mask] = v;
        dst[op + i] = v;
      }
    } else {
      const PQD_GLOBAL uint8_t* src = in.base + from;
#pragma unroll 4
      for (uint32_t i = lane; i < len; i += 64u) {
        const uint8_t v = src[i];
        ring[(op + i) & ring_mask] = v;
        dst[op + i] = v;
      }
    }
  }
  // `from_memory`: the caller knows that the ring no longer holds the source bytes (offset of 64 or more)
  __device__ __forceinline__ void match(uint32_t op, uint32_t offset, uint32_t len, bool from_memory = false) {
    // lane mod offset, for a match that repeats the `offset` < 64 bytes in front of it (a shorter one reads below its
    // own first byte anyway): six conditional subtractions
    uint32_t lm = lane;
    if (offset < 64u && len > offset) {
#pragma unroll
      for (int k = 5; k >= 0; --k) {
        const uint32_t t = offset << k;
        lm = lm >= t ? lm - t : lm;
      }
    }
    if (offset <= ring_mask + 1u && !from_memory) {
      for (uint32_t b = 0; b < len; b += 64u) {
        if (b + lane < len) {
          const uint8_t v = ring[(op + b + lm - offset) & ring_mask];
          ring[(op + b + lane) & ring_mask] = v;
          dst[op + b + lane] = v;
        }
      }
    } else {
      // The bytes were stored by this wave, long ago: its stores are waited for, and the loads read L2 (they
      // pass the vector L1, which may hold an older copy of a line).  No wider fence: nobody else is involved.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (uint32_t b = 0; b < len; b += 64u) {
        if (len > offset) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // a match that reads its own output
        if (b + lane < len) {
          const uint8_t v = __hip_atomic_load(dst + (op + b + lane - offset), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ring[(op + b + lane) & ring_mask] = v;
          dst[op + b + lane] = v;
        }
      }
    }
  }
};

}  // namespace pqd
