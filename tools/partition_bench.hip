// Exploration microbenchmark (not part of the product): out[i] = dict[id[i]] with the dictionary
// PARTITIONED over workgroups instead of streamed through them.  Workgroup b of a group of G
// (b, b+8, b+16, ... land on the same XCD) keeps part b of the dictionary in LDS for good, reads
// the ids of every chunk of its group and stores only the values whose ids fall into its part
// (8-/4-byte stores, partial lines that the G workgroups complete between them).
//   hipcc --offload-arch=gfx950 -O3 -o partition_bench partition_bench.hip && ./partition_bench
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

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 1024, R = 20, CHUNK = THREADS * R;  // 20480 values per chunk
constexpr uint32_t PART_BYTES = 128 * 1024;

// G parts; grid = 256 workgroups = 256 / G groups; group q = (b % 8) + 8 * (b / (8 * G)), part = (b / 8) % G
template <typename T, int G>
__global__ __launch_bounds__(THREADS) void k_part(const uint16_t* __restrict__ ids, const T* __restrict__ dict,
                                                  T* __restrict__ out, uint32_t n_chunks) {
  extern __shared__ u32x4 lds4[];
  T* part = reinterpret_cast<T*>(lds4);
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  const uint32_t part_id = (b / 8) % G, group = (b % 8) + 8 * (b / (8 * G)), n_groups = gridDim.x / G;
  constexpr uint32_t PER = PART_BYTES / sizeof(T);
  for (uint32_t i = tid; i < PART_BYTES / 16; i += THREADS)
    lds4[i] = reinterpret_cast<const u32x4*>(dict + (size_t)part_id * PER)[i];
  __syncthreads();
  const uint32_t base = part_id * PER;
  for (uint32_t c = group; c < n_chunks; c += n_groups) {
    const size_t v0 = (size_t)c * CHUNK;
    u32x2 p[R / 4];
#pragma unroll
    for (int k = 0; k < R / 4; ++k) p[k] = *reinterpret_cast<const u32x2*>(ids + v0 + 4 * ((size_t)k * THREADS + tid));
#pragma unroll
    for (int k = 0; k < R / 4; ++k) {
      const uint32_t id[4] = {p[k].x & 0xFFFF, p[k].x >> 16, p[k].y & 0xFFFF, p[k].y >> 16};
      T* o = out + v0 + 4 * ((size_t)k * THREADS + tid);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t a = id[j] - base;
        if (a < PER) o[j] = part[a];
      }
    }
  }
}

template <typename T, int G>
void run(const char* name, const uint16_t* d_ids, const T* d_dict, T* d_out, size_t n, const std::vector<uint16_t>& ids,
         const std::vector<T>& dict) {
  auto k = k_part<T, G>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, PART_BYTES));
  const uint32_t n_chunks = (uint32_t)(n / CHUNK);
  CHECK(hipMemset(d_out, 0, n * sizeof(T)));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(k, dim3(256), dim3(THREADS), PART_BYTES, 0, d_ids, d_dict, d_out, n_chunks);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(THREADS), PART_BYTES, 0, d_ids, d_dict, d_out, n_chunks);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  ms /= 10;
  std::vector<T> h(1 << 20);
  CHECK(hipMemcpy(h.data(), d_out + (n / 2), h.size() * sizeof(T), hipMemcpyDeviceToHost));
  bool ok = true;
  for (size_t i = 0; i < h.size(); ++i) ok = ok && h[i] == dict[ids[n / 2 + i]];
  const double alg = (double)n * (2 + sizeof(T));
  printf("%-28s %8.1f us  %6.1f Gval/s  %7.1f GB/s alg (%4.1f%% of 8 TB/s)  ok=%d\n", name, ms * 1e3, n / ms / 1e6, alg / ms / 1e6,
         alg / ms / 1e6 / 80.0, (int)ok);
}

int main() {
  const size_t n = (size_t)4880 * CHUNK;  // ~100 M
  std::vector<uint16_t> ids(n);
  uint64_t x = 88172645463325252ull;
  for (size_t i = 0; i < n; ++i) {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    ids[i] = (uint16_t)(x >> 20);
  }
  std::vector<uint64_t> d64(65536);
  std::vector<uint32_t> d32(65536);
  for (int i = 0; i < 65536; ++i) {
    d64[i] = 0x9E3779B97F4A7C15ull * (i + 1);
    d32[i] = 0x9E3779B9u * (i + 1);
  }
  uint16_t* d_ids;
  uint64_t *d_d64, *d_o64;
  uint32_t *d_d32, *d_o32;
  CHECK(hipMalloc(&d_ids, n * 2));
  CHECK(hipMalloc(&d_d64, 65536 * 8));
  CHECK(hipMalloc(&d_d32, 65536 * 4));
  CHECK(hipMalloc(&d_o64, n * 8));
  CHECK(hipMalloc(&d_o32, n * 4));
  CHECK(hipMemcpy(d_ids, ids.data(), n * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_d64, d64.data(), 65536 * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_d32, d32.data(), 65536 * 4, hipMemcpyHostToDevice));
  run<uint64_t, 4>("INT64, 4 parts of 128 KiB", d_ids, d_d64, d_o64, n, ids, d64);
  run<uint32_t, 2>("INT32, 2 parts of 128 KiB", d_ids, d_d32, d_o32, n, ids, d32);
  return 0;
}
