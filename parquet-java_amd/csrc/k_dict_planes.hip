// Plane copies of 8-byte dictionaries (pqd_rle_plan.hpp, DictPlanes): for every slot the plan kernel
// claimed in this call, the low dwords of the dictionary's entries are written to the slot's first
// plane and the high dwords to its second one.  The dictionary itself is what
// PlainValuesDictionary.PlainLongDictionary / PlainDoubleDictionary hold (parquet-column/.../values/
// dictionary/PlainValuesDictionary.java:139-237: dictionarySize little-endian 8-byte values); the
// planes are a layout of the same values that lets the gather kernel sweep them through LDS a
// quarter of a plane at a time and look an id up with one 4-byte LDS read per plane.
// Traffic: 16 bytes per entry, once per dictionary and call (a 65536-entry dictionary: 1 MiB).
#include "pqd_device.hpp"
#include "pqd_kernels.hpp"
#include "pqd_rle_plan.hpp"

namespace pqd {

constexpr int DP_THREADS = 256;

__global__ __launch_bounds__(DP_THREADS) void k_dict_planes(DictPlanes planes, const uint32_t* __restrict__ cursor) {
  uint32_t n = cursor[PLAN_CTR_PLANES];
  if (n > planes.n_slots) n = planes.n_slots;
  const uint32_t items = n * DP_PIECES;
  for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) dict_plane_piece(planes, item, threadIdx.x, DP_THREADS);
}

hipError_t launch_dict_planes(const DictPlanes& planes, const uint32_t* cursor, hipStream_t stream) {
  if (planes.n_slots == 0) return hipSuccess;
  // the number of dictionaries is known to the device only: a fixed grid strides over the work
  const uint32_t grid = planes.n_slots * DP_PIECES < 1024u ? planes.n_slots * DP_PIECES : 1024u;
  hipLaunchKernelGGL(k_dict_planes, dim3(grid), dim3(DP_THREADS), 0, stream, planes, cursor);
  return hipGetLastError();
}

}  // namespace pqd
