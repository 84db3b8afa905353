// Exploration microbenchmark (not part of the product): out[i] = dict[id[i]] for a 65536-entry
// dictionary of 8-byte values that does not fit LDS, ids as uint16 in HBM (100 M of them).
//   A. "slices": the dictionary streams through LDS in 64 KiB slices of whole entries; a pass tests
//      every id against the slice's range (subtract, compare, conditional 8-byte LDS read).
//   B. "planes": the dictionary is stored plane-major (plane p = bytes 2p, 2p+1 of every entry, i.e.
//      four arrays of 65536 x 2 bytes); a pass brings half a plane (32768 entries x 2 bytes = 64 KiB)
//      into the LDS buffer of that half, so the LDS address of an entry is always 2 * id, and a lookup
//      is ONE 2-byte LDS read into the right half of a result register under a lane mask that was
//      computed once per value (id < 32768): no subtract, no compare, no branch per pass.
// Both sweep in alternating order (the two passes a batch ends with are the two the next one starts
// with), double-buffer the LDS-DMA the same way and prefetch the next batch's ids.
//   hipcc --offload-arch=gfx950 -O3 -o plane_bench plane_bench.hip && ./plane_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define LDSP __attribute__((address_space(3)))
#define GLBP __attribute__((address_space(1)))

constexpr int THREADS = 1024;
constexpr int WAVES = THREADS / 64;
constexpr uint32_t BUF = 65536;

__device__ __forceinline__ uint32_t uni(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ __forceinline__ void dma16(const void* gbase, uint32_t goff, uint32_t lds_base) {
  const uint64_t g = (uint64_t)(uintptr_t)gbase;
  const uint64_t gs = ((uint64_t)uni((uint32_t)(g >> 32)) << 32) | uni((uint32_t)g);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base), "v"(goff), "s"(gs) : "memory");
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wg_barrier_keep() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// 64 KiB from src into the LDS buffer at byte address lds_buf, all 16 waves
__device__ __forceinline__ void load_64k(const uint8_t* src, uint32_t lds_buf, uint32_t wave, uint32_t lane) {
#pragma unroll
  for (int i = 0; i < (int)(BUF / 16 / THREADS); ++i) {
    const uint32_t piece0 = ((uint32_t)i * WAVES + wave) * 64;
    dma16(src + (size_t)piece0 * 16, lane * 16, lds_buf + piece0 * 16);
  }
}

__device__ __forceinline__ void store_nt(uint64_t* o, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  u32x4 v = {a, b, c, d};
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(o));
}

// ---------------------------------------------------------------------------------------------
// C. dword planes in quarters.  Plane 0 = the low dwords of all entries (256 KiB), plane 1 = the high
// dwords.  A pass brings one quarter of a plane (16384 entries x 4 bytes = 64 KiB) into LDS buffer
// (quarter & 1), so an entry's LDS address is always (4 * id) & 0x1FFFF.  Which quarter an id belongs to
// is kept as two lane masks per value (bit 15 and bit 14 of the id), computed once per batch; a
// lookup is one scalar instruction that puts the quarter's lanes into EXEC and one 4-byte LDS read
// into the whole result register: no vector arithmetic, no branch, no merge.
template <int PL, int QT>
__device__ __forceinline__ void lookup(uint32_t& vlo, uint32_t& vhi, uint32_t addr, uint64_t hi, uint64_t lo) {
#define PQD_LOOKUP(SETEXEC, DST)                                                                     \
  asm volatile(SETEXEC "\n\tds_read_b32 %0, %1\n\ts_mov_b64 exec, -1" : "+v"(DST) : "v"(addr), "s"(hi), "s"(lo) : "scc")
  if (PL == 0) {
    if (QT == 0) PQD_LOOKUP("s_nor_b64 exec, %2, %3", vlo);
    if (QT == 1) PQD_LOOKUP("s_andn2_b64 exec, %3, %2", vlo);
    if (QT == 2) PQD_LOOKUP("s_andn2_b64 exec, %2, %3", vlo);
    if (QT == 3) PQD_LOOKUP("s_and_b64 exec, %2, %3", vlo);
  } else {
    if (QT == 0) PQD_LOOKUP("s_nor_b64 exec, %2, %3", vhi);
    if (QT == 1) PQD_LOOKUP("s_andn2_b64 exec, %3, %2", vhi);
    if (QT == 2) PQD_LOOKUP("s_andn2_b64 exec, %2, %3", vhi);
    if (QT == 3) PQD_LOOKUP("s_and_b64 exec, %2, %3", vhi);
  }
#undef PQD_LOOKUP
}

// Pass Q of a sweep (Q = 0..7 in sweep order; DESC: the sweep runs from the last quarter down).
// SKEL 1: no lookups; SKEL 2: no DMA after the first two quarters (timing only, wrong results).
template <int R, int Q, bool DESC, int SKEL>
__device__ __forceinline__ void plane_pass(const uint8_t* planes, uint32_t lds0, uint32_t wave, uint32_t lane,
                                           uint32_t (&vlo)[R], uint32_t (&vhi)[R], const uint32_t (&addr)[R],
                                           const uint64_t (&mhi)[R], const uint64_t (&mlo)[R]) {
  constexpr int S = DESC ? 7 - Q : Q;  // plane S >> 2, quarter S & 3
  constexpr int PL = S >> 2, QT = S & 3;
  if (Q == 0)
    wg_barrier_keep<R / 4>();  // the id prefetch stays in flight
  else
    wg_barrier();
  if (Q >= 1 && Q + 1 < 8 && SKEL != 2) {  // passes 0 and 1 find their quarters resident
    constexpr int NX = DESC ? S - 1 : S + 1;
    load_64k(planes + (size_t)NX * BUF, lds0 + (uint32_t)(NX & 1) * BUF, wave, lane);
  }
  if (SKEL == 1 || SKEL == 3) return;
#pragma unroll
  for (int r = 0; r < R; ++r) lookup<PL, QT>(vlo[r], vhi[r], addr[r], mhi[r], mlo[r]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Early refill: the quarter of pass Q + 2 is requested right behind the lookups of pass Q (after a barrier for the
// LDS reads alone), before the wait for pass Q + 1's quarter: two transfers in flight instead of one.
template <int R, int Q, bool DESC, int SKEL>
__device__ __forceinline__ void plane_pass_early(const uint8_t* planes, uint32_t lds0, uint32_t wave, uint32_t lane,
                                                 uint32_t (&vlo)[R], uint32_t (&vhi)[R], const uint32_t (&addr)[R],
                                                 const uint64_t (&mhi)[R], const uint64_t (&mlo)[R]) {
  constexpr int S = DESC ? 7 - Q : Q;
  constexpr int PL = S >> 2, QT = S & 3;
  if (Q >= 2) {  // this pass's quarter has landed (the one behind it may still be on its way)
    if (Q < 7)
      asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (SKEL != 1 && SKEL != 3) {
#pragma unroll
    for (int r = 0; r < R; ++r) lookup<PL, QT>(vlo[r], vhi[r], addr[r], mhi[r], mlo[r]);
  }
  if (Q + 2 < 8) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everybody is done reading this buffer
    if (SKEL != 2) {
      constexpr int NX = DESC ? S - 2 : S + 2;
      load_64k(planes + (size_t)NX * BUF, lds0 + (uint32_t)(NX & 1) * BUF, wave, lane);
    }
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}
template <int R, bool DESC, int SKEL>
__device__ __forceinline__ void plane_sweep_early(const uint8_t* planes, uint32_t lds0, uint32_t wave, uint32_t lane,
                                            uint32_t (&vlo)[R], uint32_t (&vhi)[R], const uint32_t (&addr)[R],
                                            const uint64_t (&mhi)[R], const uint64_t (&mlo)[R]) {
  plane_pass_early<R, 0, DESC, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
  plane_pass_early<R, 1, DESC, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
  plane_pass_early<R, 2, DESC, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
  plane_pass_early<R, 3, DESC, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
  plane_pass_early<R, 4, DESC, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
  plane_pass_early<R, 5, DESC, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
  plane_pass_early<R, 6, DESC, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
  plane_pass_early<R, 7, DESC, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
}

template <int R, bool DESC, int SKEL>
__device__ __forceinline__ void plane_sweep(const uint8_t* planes, uint32_t lds0, uint32_t wave, uint32_t lane,
                                            uint32_t (&vlo)[R], uint32_t (&vhi)[R], const uint32_t (&addr)[R],
                                            const uint64_t (&mhi)[R], const uint64_t (&mlo)[R]) {
  plane_pass<R, 0, DESC, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
  plane_pass<R, 1, DESC, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
  plane_pass<R, 2, DESC, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
  plane_pass<R, 3, DESC, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
  plane_pass<R, 4, DESC, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
  plane_pass<R, 5, DESC, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
  plane_pass<R, 6, DESC, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
  plane_pass<R, 7, DESC, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
}

template <int R, int SKEL, int SCHED = 0>
__global__ __launch_bounds__(THREADS) void k_planes(const uint16_t* __restrict__ ids, const uint8_t* __restrict__ planes,
                                                    uint64_t* __restrict__ out, size_t n) {
  extern __shared__ u32x4 lds4[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDSP uint8_t*)lds4;
  constexpr int UNITS = R / 4;
  const uint32_t tid = threadIdx.x, wave = uni(tid >> 6), lane = tid & 63u;
  const size_t batch = (size_t)THREADS * R;
  bool descending = false;
  load_64k(planes, lds0, wave, lane);
  load_64k(planes + BUF, lds0 + BUF, wave, lane);
  u32x2 nxt[UNITS];
  size_t b0 = (size_t)blockIdx.x * batch;
#pragma unroll
  for (int k = 0; k < UNITS; ++k) {
    const size_t u = b0 / 4 + (size_t)k * THREADS + tid;
    nxt[k] = u * 4 < n ? *reinterpret_cast<const u32x2*>(ids + u * 4) : u32x2{0, 0};
  }
  wg_barrier();
  for (; b0 < n; b0 += (size_t)gridDim.x * batch) {
    uint32_t addr[R];
    uint64_t mhi[R], mlo[R];
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
      addr[4 * k] = (nxt[k].x & 0xFFFF) * 4;
      addr[4 * k + 1] = (nxt[k].x >> 16) * 4;
      addr[4 * k + 2] = (nxt[k].y & 0xFFFF) * 4;
      addr[4 * k + 3] = (nxt[k].y >> 16) * 4;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      mhi[r] = __builtin_amdgcn_ballot_w64((addr[r] & 0x20000u) != 0);
      mlo[r] = __builtin_amdgcn_ballot_w64((addr[r] & 0x10000u) != 0);
      addr[r] &= 0x1FFFFu;
    }
    const size_t nb0 = b0 + (size_t)gridDim.x * batch;
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
      const size_t u = nb0 / 4 + (size_t)k * THREADS + tid;
      nxt[k] = (SKEL != 3 && u * 4 < n) ? *reinterpret_cast<const u32x2*>(ids + u * 4) : u32x2{0, 0};
    }
    uint32_t vlo[R], vhi[R];
#pragma unroll
    for (int r = 0; r < R; ++r) vlo[r] = vhi[r] = 0;
    if (SCHED == 0) {
      if (!descending)
        plane_sweep<R, false, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
      else
        plane_sweep<R, true, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
    } else {
      if (!descending)
        plane_sweep_early<R, false, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
      else
        plane_sweep_early<R, true, SKEL>(planes, lds0, wave, lane, vlo, vhi, addr, mhi, mlo);
    }
    descending = !descending;
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
      const size_t u = b0 / 4 + (size_t)k * THREADS + tid;
      if (SKEL != 3 && u * 4 < n) {
        store_nt(out + u * 4, vlo[4 * k], vhi[4 * k], vlo[4 * k + 1], vhi[4 * k + 1]);
        store_nt(out + u * 4 + 2, vlo[4 * k + 2], vhi[4 * k + 2], vlo[4 * k + 3], vhi[4 * k + 3]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// D. the product's form of the lookup: the quarter of every id kept as a byte (four values per register), a
// byte-selecting compare puts the quarter's lanes into EXEC (one VALU instruction per lookup, no scalar masks).
template <int J>
__device__ __forceinline__ void tag_lookup(uint32_t& dst, uint32_t addr, uint32_t tags, uint32_t quarter, uint64_t all) {
#define PQD_TAG_LOOKUP(SEL)                                                                              \
  asm volatile("v_cmpx_eq_u32_sdwa vcc, %2, %3 src0_sel:" SEL " src1_sel:DWORD\n\tds_read_b32 %0, %1\n\ts_mov_b64 exec, %4" \
               : "+v"(dst)                                                                                 \
               : "v"(addr), "v"(tags), "s"(quarter), "s"(all)                                              \
               : "vcc")
  if (J == 0) PQD_TAG_LOOKUP("BYTE_0");
  if (J == 1) PQD_TAG_LOOKUP("BYTE_1");
  if (J == 2) PQD_TAG_LOOKUP("BYTE_2");
  if (J == 3) PQD_TAG_LOOKUP("BYTE_3");
#undef PQD_TAG_LOOKUP
}

template <int R, int SHARE>
__global__ __launch_bounds__(THREADS) void k_planes_tags(const uint16_t* __restrict__ ids, const uint8_t* __restrict__ planes,
                                                         uint64_t* __restrict__ out, size_t n) {
  extern __shared__ u32x4 lds4[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDSP uint8_t*)lds4;
  constexpr int UNITS = R / 4;
  const uint32_t tid = threadIdx.x, wave = uni(tid >> 6), lane = tid & 63u;
  const size_t batch = (size_t)THREADS * R;
  const size_t n_batches = (n + batch - 1) / batch;
  size_t bi, bi_end, bi_step;
  if (SHARE) {  // contiguous shares, as the product
    const size_t share = (n_batches + gridDim.x - 1) / gridDim.x;
    bi = blockIdx.x * share;
    bi_end = bi + share < n_batches ? bi + share : n_batches;
    bi_step = 1;
  } else {
    bi = blockIdx.x;
    bi_end = n_batches;
    bi_step = gridDim.x;
  }
  if (bi >= bi_end) return;
  bool descending = false;
  load_64k(planes, lds0, wave, lane);
  load_64k(planes + BUF, lds0 + BUF, wave, lane);
  u32x2 nxt[UNITS];
#pragma unroll
  for (int k = 0; k < UNITS; ++k) {
    const size_t u = bi * batch / 4 + (size_t)k * THREADS + tid;
    nxt[k] = u * 4 < n ? *reinterpret_cast<const u32x2*>(ids + u * 4) : u32x2{0, 0};
  }
  wg_barrier();
  const uint64_t all = __builtin_amdgcn_ballot_w64(true);
  for (; bi < bi_end; bi += bi_step) {
    const size_t b0 = bi * batch;
    uint32_t addr[R], tags[UNITS];
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
      const uint32_t i0 = nxt[k].x & 0xFFFF, i1 = nxt[k].x >> 16, i2 = nxt[k].y & 0xFFFF, i3 = nxt[k].y >> 16;
      tags[k] = (i0 >> 14) | ((i1 >> 14) << 8) | ((i2 >> 14) << 16) | ((i3 >> 14) << 24);
      addr[4 * k] = (i0 << 2) & 0x1FFFFu;
      addr[4 * k + 1] = (i1 << 2) & 0x1FFFFu;
      addr[4 * k + 2] = (i2 << 2) & 0x1FFFFu;
      addr[4 * k + 3] = (i3 << 2) & 0x1FFFFu;
    }
    const size_t nb = bi + bi_step;
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
      const size_t u = nb * batch / 4 + (size_t)k * THREADS + tid;
      nxt[k] = (nb < bi_end && u * 4 < n) ? *reinterpret_cast<const u32x2*>(ids + u * 4) : u32x2{0, 0};
    }
    uint32_t vlo[R], vhi[R];
#pragma unroll
    for (int r = 0; r < R; ++r) vlo[r] = vhi[r] = 0;
    for (uint32_t q = 0; q < 8; ++q) {
      const uint32_t s = descending ? 7 - q : q;
      if (q == 0)
        wg_barrier_keep<R / 4>();
      else
        wg_barrier();
      if (q >= 1 && q + 1 < 8) {
        const uint32_t nx = descending ? s - 1 : s + 1;
        load_64k(planes + (size_t)nx * BUF, lds0 + (nx & 1u) * BUF, wave, lane);
      }
      const uint32_t qt = uni(s & 3u);
      if (s < 4) {
#pragma unroll
        for (int k = 0; k < UNITS; ++k) {
          tag_lookup<0>(vlo[4 * k], addr[4 * k], tags[k], qt, all);
          tag_lookup<1>(vlo[4 * k + 1], addr[4 * k + 1], tags[k], qt, all);
          tag_lookup<2>(vlo[4 * k + 2], addr[4 * k + 2], tags[k], qt, all);
          tag_lookup<3>(vlo[4 * k + 3], addr[4 * k + 3], tags[k], qt, all);
        }
      } else {
#pragma unroll
        for (int k = 0; k < UNITS; ++k) {
          tag_lookup<0>(vhi[4 * k], addr[4 * k], tags[k], qt, all);
          tag_lookup<1>(vhi[4 * k + 1], addr[4 * k + 1], tags[k], qt, all);
          tag_lookup<2>(vhi[4 * k + 2], addr[4 * k + 2], tags[k], qt, all);
          tag_lookup<3>(vhi[4 * k + 3], addr[4 * k + 3], tags[k], qt, all);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    descending = !descending;
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
      const size_t u = b0 / 4 + (size_t)k * THREADS + tid;
      if (u * 4 < n) {
        store_nt(out + u * 4, vlo[4 * k], vhi[4 * k], vlo[4 * k + 1], vhi[4 * k + 1]);
        store_nt(out + u * 4 + 2, vlo[4 * k + 2], vhi[4 * k + 2], vlo[4 * k + 3], vhi[4 * k + 3]);
      }
    }
  }
}

template <typename F>
float time_it(F f, int iters = 10) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  f();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) f();
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / iters * 1000.f;
}

int main() {
  const size_t n = 100000000;
  const uint32_t D = 65536;
  std::vector<uint16_t> h_ids(n);
  uint64_t x = 88172645463325252ull;
  for (size_t i = 0; i < n; ++i) {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    h_ids[i] = (uint16_t)(x >> 20);
  }
  std::vector<uint64_t> h_dict(D);
  for (uint32_t i = 0; i < D; ++i) {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    h_dict[i] = x;
  }
  std::vector<uint32_t> h_planes((size_t)2 * D);
  for (uint32_t e = 0; e < D; ++e) {
    h_planes[e] = (uint32_t)h_dict[e];
    h_planes[(size_t)D + e] = (uint32_t)(h_dict[e] >> 32);
  }
  uint16_t* ids;
  uint64_t *dict, *out;
  uint8_t* planes;
  CHECK(hipMalloc(&ids, n * 2 + 64));
  CHECK(hipMalloc(&dict, D * 8));
  CHECK(hipMalloc(&planes, D * 8));
  CHECK(hipMalloc(&out, n * 8));
  CHECK(hipMemcpy(ids, h_ids.data(), n * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dict, h_dict.data(), D * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(planes, h_planes.data(), D * 8, hipMemcpyHostToDevice));
  std::vector<uint64_t> h_out(1 << 16);
  auto check = [&](const char* name) {
    for (size_t at : {(size_t)0, n / 2, n - h_out.size()}) {
      CHECK(hipMemcpy(h_out.data(), out + at, h_out.size() * 8, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < h_out.size(); ++i)
        if (h_out[i] != h_dict[h_ids[at + i]]) {
          printf("  %s MISMATCH at %zu: %016llx vs %016llx\n", name, at + i, (unsigned long long)h_out[i],
                 (unsigned long long)h_dict[h_ids[at + i]]);
          return;
        }
    }
    CHECK(hipMemset(out, 0, n * 8));
  };
  const double bytes = (double)n * 10;
  auto report = [&](const char* name, float us) {
    printf("  %-52s %8.1f us  %7.1f GB/s  %5.1f %% of 8 TB/s\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 80.0);
    fflush(stdout);
  };
  float us;
#define RUN_PLANES(R, SKEL) RUN_PLANES_S(R, SKEL, 0)
#define RUN_PLANES_S(R, SKEL, SCHED)                                                                                                \
  {                                                                                                                            \
    CHECK(hipFuncSetAttribute((const void*)k_planes<R, SKEL, SCHED>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));           \
    us = time_it([&] { hipLaunchKernelGGL((k_planes<R, SKEL, SCHED>), dim3(256), dim3(THREADS), 2 * BUF, 0, ids, planes, out, n); }); \
    if (!SKEL) check("planes"); else CHECK(hipMemset(out, 0, n * 8));                                                                                                \
    char nm[96];                                                                                                               \
    snprintf(nm, sizeof nm, "dword planes in quarters  R=%d%s%s", R, SCHED ? " early" : "", SKEL == 3 ? "  (DMA only)" : SKEL == 1 ? "  (no lookups: DMA + ids + stores)" : SKEL == 2 ? "  (no DMA: ids + lookups + stores)" : "");                           \
    report(nm, us);                                                                                                            \
  }
#define RUN_TAGS(R, SHARE)                                                                                                     \
  {                                                                                                                            \
    CHECK(hipFuncSetAttribute((const void*)k_planes_tags<R, SHARE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));     \
    us = time_it([&] { hipLaunchKernelGGL((k_planes_tags<R, SHARE>), dim3(256), dim3(THREADS), 2 * BUF, 0, ids, planes, out, n); }); \
    check("tags");                                                                                                             \
    report(SHARE ? "byte tags + cmpx, contiguous shares" : "byte tags + cmpx, grid stride", us);                               \
  }
  RUN_PLANES(20, 0);
  RUN_TAGS(20, 0);
  RUN_TAGS(20, 1);
  RUN_PLANES(20, 0);
  RUN_TAGS(20, 0);
  RUN_TAGS(20, 1);
  return 0;
  RUN_PLANES(20, 1);
  RUN_PLANES(20, 2);
  RUN_PLANES(20, 3);
  RUN_PLANES_S(20, 0, 1);
  RUN_PLANES_S(20, 1, 1);
  RUN_PLANES_S(20, 2, 1);
  RUN_PLANES_S(20, 3, 1);
  RUN_PLANES(20, 0);
  RUN_PLANES_S(20, 0, 1);
  return 0;
}
