// Exploration microbenchmark (not part of the product): strategies for out[i] = dict[id[i]] with a
// dictionary larger than LDS.  ids are uint16 in HBM (100 M), dictionary 65536 x {4,8} bytes.
//   hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip && ./gather_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e = (x);                                                               \
    if (e != hipSuccess) {                                                            \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);    \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---- K0: direct gather, 4 ids per thread per step
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_direct(const uint16_t* __restrict__ ids, const T* __restrict__ dict,
                                                T* __restrict__ out, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < n / 4; u += stride) {
    u32x2 p = *reinterpret_cast<const u32x2*>(ids + u * 4);
    uint32_t id[4] = {p.x & 0xFFFF, p.x >> 16, p.y & 0xFFFF, p.y >> 16};
    T v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE == 0) v[j] = dict[id[j]];
      if (MODE == 1) v[j] = __builtin_nontemporal_load(dict + id[j]);
      if (MODE == 2) v[j] = __hip_atomic_load(dict + id[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) out[u * 4 + j] = v[j];
  }
}

// ---- K1: multi-pass LDS slices.  WG of THREADS threads holds R ids per thread in registers; the
// dictionary is streamed through LDS in slices of S entries; each pass resolves the ids that fall
// into the slice.  Results are stored once, coalesced.
template <typename T, int THREADS, int R, int S, bool PRED>
__global__ __launch_bounds__(THREADS) void k_slices(const uint16_t* __restrict__ ids, const T* __restrict__ dict,
                                                    uint32_t dict_len, T* __restrict__ out, size_t n) {
  extern __shared__ u32x4 lds4[];
  T* lds = reinterpret_cast<T*>(lds4);
  const size_t batch = (size_t)THREADS * R;
  for (size_t b0 = (size_t)blockIdx.x * batch; b0 < n; b0 += (size_t)gridDim.x * batch) {
    // each thread owns R/4 units of 4 consecutive values, units strided by THREADS
    uint32_t id[R];
    T v[R];
#pragma unroll
    for (int k = 0; k < R / 4; ++k) {
      size_t u = b0 / 4 + (size_t)k * THREADS + threadIdx.x;
      u32x2 p = u * 4 < n ? *reinterpret_cast<const u32x2*>(ids + u * 4) : u32x2{0, 0};
      id[4 * k] = p.x & 0xFFFF;
      id[4 * k + 1] = p.x >> 16;
      id[4 * k + 2] = p.y & 0xFFFF;
      id[4 * k + 3] = p.y >> 16;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = 0;
    for (uint32_t s0 = 0; s0 < dict_len; s0 += S) {
      __syncthreads();
      // stage dict[s0 .. s0+S) into LDS, 16 bytes per lane
      const u32x4* src = reinterpret_cast<const u32x4*>(dict + s0);
      for (int i = threadIdx.x; i < (int)(S * sizeof(T) / 16); i += THREADS) lds4[i] = src[i];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < R; ++r) {
        uint32_t idx = id[r] - s0;
        if (PRED) {
          if (idx < (uint32_t)S) v[r] = lds[idx];
        } else {
          T x = lds[idx & (S - 1)];
          v[r] = idx < (uint32_t)S ? x : v[r];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < R / 4; ++k) {
      size_t u = b0 / 4 + (size_t)k * THREADS + threadIdx.x;
      if (u * 4 < n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[u * 4 + j] = v[4 * k + j];
      }
    }
  }
}


// ---- K3: K1 with double-buffered slices staged by LDS-DMA (global_load_lds_dwordx4), one barrier per pass
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

template <typename T, int THREADS, int S>
__device__ __forceinline__ void prefetch_slice(const T* dict, uint32_t s0, T* buf) {
  // S*sizeof(T)/16 pieces of 16 bytes; a wave instruction moves 64 pieces (1 KiB)
  const int pieces = S * sizeof(T) / 16;
  const char* src = reinterpret_cast<const char*>(dict + s0);
  char* dst = reinterpret_cast<char*>(buf);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < pieces / THREADS; ++i) {
    const int piece0 = i * THREADS + wave * 64;  // wave-uniform
    __builtin_amdgcn_global_load_lds((gbl_void*)(src + (size_t)(piece0 + lane) * 16), (lds_void*)(dst + (size_t)piece0 * 16),
                                     16, 0, 0);
  }
}

template <typename T, int THREADS, int R, int S, int ROT = 0, int SKEL = 0>
__global__ __launch_bounds__(THREADS) void k_slices_db(const uint16_t* __restrict__ ids, const T* __restrict__ dict,
                                                       uint32_t dict_len, T* __restrict__ out, size_t n) {
  extern __shared__ u32x4 lds4[];
  T* lds = reinterpret_cast<T*>(lds4);
  const size_t batch = (size_t)THREADS * R;
  const uint32_t P = dict_len / S;
  uint32_t q = 0;  // pass counter; buffer q&1 holds slice q%P
  const uint32_t rot = ROT ? (blockIdx.x * ROT) % P : 0;
  prefetch_slice<T, THREADS, S>(dict, rot * S, lds);
  for (size_t b0 = (size_t)blockIdx.x * batch; b0 < n; b0 += (size_t)gridDim.x * batch) {
    uint32_t id[R];
    T v[R];
#pragma unroll
    for (int k = 0; k < R / 4; ++k) {
      size_t u = b0 / 4 + (size_t)k * THREADS + threadIdx.x;
      u32x2 p = u * 4 < n ? *reinterpret_cast<const u32x2*>(ids + u * 4) : u32x2{0, 0};
      id[4 * k] = p.x & 0xFFFF;
      id[4 * k + 1] = p.x >> 16;
      id[4 * k + 2] = p.y & 0xFFFF;
      id[4 * k + 3] = p.y >> 16;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = 0;
    for (uint32_t s = 0; s < P; ++s, ++q) {
      __syncthreads();  // slice q landed (vmcnt(0) + barrier); everyone is done with the other buffer
      const uint32_t sl = (s + rot) % P;
      const uint32_t nxt = (sl + 1 == P) ? 0 : sl + 1;
      prefetch_slice<T, THREADS, S>(dict, nxt * S, lds + (size_t)((q + 1) & 1) * S);
      const T* cur = lds + (size_t)(q & 1) * S;
      const uint32_t s0 = sl * S;
      if (SKEL) continue;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        uint32_t idx = id[r] - s0;
        if (idx < (uint32_t)S) v[r] = cur[idx];
      }
    }
#pragma unroll
    for (int k = 0; k < R / 4; ++k) {
      size_t u = b0 / 4 + (size_t)k * THREADS + threadIdx.x;
      if (u * 4 < n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[u * 4 + j] = v[4 * k + j];
      }
    }
  }
}


// ---- K4: K3 with ids kept packed (two 16-bit ids per register)
template <typename T, int THREADS, int R, int S>
__global__ __launch_bounds__(THREADS) void k_slices_dbp(const uint16_t* __restrict__ ids, const T* __restrict__ dict,
                                                        uint32_t dict_len, T* __restrict__ out, size_t n) {
  extern __shared__ u32x4 lds4[];
  T* lds = reinterpret_cast<T*>(lds4);
  const size_t batch = (size_t)THREADS * R;
  const uint32_t P = dict_len / S;
  uint32_t q = 0;
  prefetch_slice<T, THREADS, S>(dict, 0, lds);
  for (size_t b0 = (size_t)blockIdx.x * batch; b0 < n; b0 += (size_t)gridDim.x * batch) {
    uint32_t idp[R / 2];
    T v[R];
#pragma unroll
    for (int k = 0; k < R / 4; ++k) {
      size_t u = b0 / 4 + (size_t)k * THREADS + threadIdx.x;
      u32x2 p = u * 4 < n ? *reinterpret_cast<const u32x2*>(ids + u * 4) : u32x2{0, 0};
      idp[2 * k] = p.x;
      idp[2 * k + 1] = p.y;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = 0;
    for (uint32_t s = 0; s < P; ++s, ++q) {
      __syncthreads();
      const uint32_t nxt = (s + 1 == P) ? 0 : s + 1;
      prefetch_slice<T, THREADS, S>(dict, nxt * S, lds + (size_t)((q + 1) & 1) * S);
      const T* cur = lds + (size_t)(q & 1) * S;
      const uint32_t s0 = s * S;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        uint32_t id = (r & 1) ? (idp[r >> 1] >> 16) : (idp[r >> 1] & 0xFFFF);
        uint32_t idx = id - s0;
        if (idx < (uint32_t)S) v[r] = cur[idx];
      }
    }
#pragma unroll
    for (int k = 0; k < R / 4; ++k) {
      size_t u = b0 / 4 + (size_t)k * THREADS + threadIdx.x;
      if (u * 4 < n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[u * 4 + j] = v[4 * k + j];
      }
    }
  }
}

// ---- K2: whole (small) dictionary in LDS, single pass: the ceiling of the LDS gather
template <typename T, int THREADS>
__global__ __launch_bounds__(THREADS) void k_lds_full(const uint16_t* __restrict__ ids, const T* __restrict__ dict,
                                                      uint32_t dict_len, T* __restrict__ out, size_t n) {
  extern __shared__ u32x4 lds4[];
  T* lds = reinterpret_cast<T*>(lds4);
  const u32x4* src = reinterpret_cast<const u32x4*>(dict);
  for (int i = threadIdx.x; i < (int)(dict_len * sizeof(T) / 16); i += THREADS) lds4[i] = src[i];
  __syncthreads();
  size_t stride = (size_t)gridDim.x * THREADS;
  for (size_t u = (size_t)blockIdx.x * THREADS + threadIdx.x; u < n / 4; u += stride) {
    u32x2 p = *reinterpret_cast<const u32x2*>(ids + u * 4);
    uint32_t id[4] = {p.x & 0xFFFF, p.x >> 16, p.y & 0xFFFF, p.y >> 16};
    T v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = lds[id[j] & (dict_len - 1)];
#pragma unroll
    for (int j = 0; j < 4; ++j) out[u * 4 + j] = v[j];
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

template <typename T>
void run(size_t n) {
  const uint32_t D = 65536;
  std::vector<uint16_t> h_ids(n);
  uint64_t x = 88172645463325252ull;
  for (size_t i = 0; i < n; ++i) {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    h_ids[i] = (uint16_t)(x >> 20);
  }
  std::vector<T> h_dict(D);
  for (uint32_t i = 0; i < D; ++i) h_dict[i] = (T)(i * 2654435761u + 12345);
  uint16_t* ids;
  T *dict, *out;
  CHECK(hipMalloc(&ids, n * 2));
  CHECK(hipMalloc(&dict, D * sizeof(T)));
  CHECK(hipMalloc(&out, n * sizeof(T)));
  CHECK(hipMemcpy(ids, h_ids.data(), n * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dict, h_dict.data(), D * sizeof(T), hipMemcpyHostToDevice));
  std::vector<T> h_out(4096);
  auto check = [&](const char* name) {
    CHECK(hipMemcpy(h_out.data(), out + (n - 4096), 4096 * sizeof(T), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4096; ++i)
      if (h_out[i] != h_dict[h_ids[n - 4096 + i]]) {
        printf("  %s MISMATCH at %d\n", name, i);
        return;
      }
    CHECK(hipMemset(out, 0, n * sizeof(T)));
  };
  const double bytes = (double)n * (2 + sizeof(T));
  auto report = [&](const char* name, float us) {
    printf("  T=%zu %-44s %8.1f us  %7.1f GB/s  %6.1f Gval/s\n", sizeof(T), name, us, bytes / us / 1e3, n / us / 1e3);
    fflush(stdout);
  };
  float us;
  us = time_it([&] { hipLaunchKernelGGL((k_direct<T, 0>), dim3(256 * 16), dim3(256), 0, 0, ids, dict, out, n); });
  check("direct");
  report("direct gather (plain loads)", us);
  us = time_it([&] { hipLaunchKernelGGL((k_direct<T, 1>), dim3(256 * 16), dim3(256), 0, 0, ids, dict, out, n); });
  check("direct nt");
  report("direct gather (nontemporal loads)", us);
  us = time_it([&] { hipLaunchKernelGGL((k_direct<T, 2>), dim3(256 * 16), dim3(256), 0, 0, ids, dict, out, n); });
  check("direct sc1");
  report("direct gather (agent-scope relaxed loads)", us);
  us = time_it([&] { hipLaunchKernelGGL((k_direct<T, 0>), dim3(256 * 32), dim3(256), 0, 0, ids, dict, out, n); });
  check("direct 32/CU");
  report("direct gather, grid 32 WG/CU", us);

#define SL(THREADS, R, S, PRED, WGS)                                                                            \
  {                                                                                                             \
    size_t lds = (size_t)S * sizeof(T);                                                                         \
    CHECK(hipFuncSetAttribute((const void*)k_slices<T, THREADS, R, S, PRED>,                                    \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
    us = time_it([&] {                                                                                          \
      hipLaunchKernelGGL((k_slices<T, THREADS, R, S, PRED>), dim3(256 * WGS), dim3(THREADS), lds, 0, ids, dict, \
                         D, out, n);                                                                            \
    });                                                                                                         \
    check("slices");                                                                                            \
    char nm[128];                                                                                               \
    snprintf(nm, sizeof nm, "LDS slices thr=%d R=%d S=%d pred=%d wg/cu=%d", THREADS, R, S, (int)PRED, WGS);     \
    report(nm, us);                                                                                             \
  }
  SL(1024, 32, 16384, true, 1);
  SL(1024, 32, 16384, false, 1);
  SL(1024, 16, 16384, true, 1);
  SL(512, 32, 8192, true, 2);
  SL(512, 32, 8192, false, 2);
  SL(1024, 32, 8192, true, 2);
  SL(256, 32, 4096, true, 8);
  SL(256, 32, 4096, false, 8);
  if (sizeof(T) == 4) {
    SL(1024, 32, 32768, true, 1);
    SL(1024, 32, 32768, false, 1);
  }

#define SLDBX(THREADS, R, S, ROT, SKEL)                                                                          \
  {                                                                                                              \
    size_t lds = (size_t)2 * S * sizeof(T);                                                                      \
    CHECK(hipFuncSetAttribute((const void*)k_slices_db<T, THREADS, R, S, ROT, SKEL>,                             \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                            \
    us = time_it([&] {                                                                                           \
      hipLaunchKernelGGL((k_slices_db<T, THREADS, R, S, ROT, SKEL>), dim3(256), dim3(THREADS), lds, 0, ids, dict, \
                         D, out, n);                                                                             \
    });                                                                                                          \
    if (!SKEL) check("slices dbx");                                                                              \
    char nm[128];                                                                                                \
    snprintf(nm, sizeof nm, "dbuf thr=%d R=%d S=%d rot=%d skeleton=%d", THREADS, R, S, ROT, SKEL);                \
    report(nm, us);                                                                                              \
  }
  SLDBX(1024, 20, (65536 / sizeof(T)), 0, 0);
  SLDBX(1024, 20, (65536 / sizeof(T)), 0, 1);
  SLDBX(1024, 20, (65536 / sizeof(T)), 1, 0);
  SLDBX(1024, 20, (65536 / sizeof(T)), 1, 1);
  SLDBX(1024, 20, (32768 / sizeof(T)), 1, 1);
  SLDBX(1024, 20, (32768 / sizeof(T)), 0, 1);
  SLDBX(512, 20, (65536 / sizeof(T)), 0, 1);
  SLDBX(256, 20, (65536 / sizeof(T)), 0, 1);
#define SLDB(THREADS, R, S, WGS)                                                                                 \
  {                                                                                                              \
    size_t lds = (size_t)2 * S * sizeof(T);                                                                      \
    CHECK(hipFuncSetAttribute((const void*)k_slices_db<T, THREADS, R, S>,                                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                            \
    us = time_it([&] {                                                                                           \
      hipLaunchKernelGGL((k_slices_db<T, THREADS, R, S>), dim3(256 * WGS), dim3(THREADS), lds, 0, ids, dict, D,   \
                         out, n);                                                                                \
    });                                                                                                          \
    check("slices db");                                                                                          \
    char nm[128];                                                                                                \
    snprintf(nm, sizeof nm, "LDS-DMA double-buffered thr=%d R=%d S=%d wg/cu=%d", THREADS, R, S, WGS);             \
    report(nm, us);                                                                                              \
  }
  SLDB(1024, 32, 8192, 1);
  SLDB(1024, 20, 8192, 1);
  SLDB(1024, 32, 4096, 1);
  SLDB(512, 32, 4096, 2);
  SLDB(512, 32, 8192, 1);
  if (sizeof(T) == 4) {
    SLDB(1024, 32, 16384, 1);
    SLDB(1024, 20, 16384, 1);
  }

#define SLDBP(THREADS, R, S, WGS)                                                                                \
  {                                                                                                              \
    size_t lds = (size_t)2 * S * sizeof(T);                                                                      \
    CHECK(hipFuncSetAttribute((const void*)k_slices_dbp<T, THREADS, R, S>,                                       \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                            \
    us = time_it([&] {                                                                                           \
      hipLaunchKernelGGL((k_slices_dbp<T, THREADS, R, S>), dim3(256 * WGS), dim3(THREADS), lds, 0, ids, dict, D,  \
                         out, n);                                                                                \
    });                                                                                                          \
    check("slices dbp");                                                                                         \
    char nm[128];                                                                                                \
    snprintf(nm, sizeof nm, "packed-id dbuf thr=%d R=%d S=%d wg/cu=%d", THREADS, R, S, WGS);                      \
    report(nm, us);                                                                                              \
  }
  SLDBP(1024, 32, 8192, 1);
  SLDBP(1024, 24, 8192, 1);
  SLDBP(512, 64, 8192, 1);
  SLDBP(512, 48, 8192, 1);
  SLDBP(512, 40, 8192, 1);
  SLDBP(256, 128, 8192, 1);
  if (sizeof(T) == 4) {
    SLDBP(1024, 32, 16384, 1);
    SLDBP(512, 64, 16384, 1);
    SLDBP(512, 40, 16384, 1);
  }
  // ceiling: a 16 K-entry dictionary wholly in LDS
  {
    const uint32_t small = 16384;
    size_t lds = small * sizeof(T);
    CHECK(hipFuncSetAttribute((const void*)k_lds_full<T, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    us = time_it([&] { hipLaunchKernelGGL((k_lds_full<T, 256>), dim3(256 * 2), dim3(256), lds, 0, ids, dict, small, out, n); });
    report("whole 16K-entry dict in LDS (ceiling)", us);
    CHECK(hipMemset(out, 0, n * sizeof(T)));
  }
  CHECK(hipFree(ids));
  CHECK(hipFree(dict));
  CHECK(hipFree(out));
}

int main() {
  size_t n = 100000000;
  run<uint32_t>(n);
  run<uint64_t>(n);
  return 0;
}
