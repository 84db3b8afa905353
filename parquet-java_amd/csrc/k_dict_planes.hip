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
constexpr uint32_t DP_PIECE_ENTRIES = 2048;                      // entries one workgroup moves per step
constexpr uint32_t DP_PIECES = 65536 / DP_PIECE_ENTRIES;         // pieces of the largest dictionary

__global__ __launch_bounds__(DP_THREADS) void k_dict_planes(DictPlanes planes, const uint32_t* __restrict__ cursor) {
  uint32_t n = cursor[2];
  if (n > planes.n_slots) n = planes.n_slots;
  const uint32_t items = n * DP_PIECES;
  for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) {
    const uint32_t slot = planes.list[item / DP_PIECES];
    if (slot >= planes.n_slots) continue;
    const uint64_t key = planes.keys[slot];
    const uint32_t entries = (uint32_t)(key >> 48) + 1u;
    const uint32_t e0 = (item % DP_PIECES) * DP_PIECE_ENTRIES;
    if (e0 >= entries) continue;
    const uint32_t* src = reinterpret_cast<const uint32_t*>((uintptr_t)(key & 0xFFFFFFFFFFFFull));  // 16-byte aligned (plan kernel)
    uint32_t* lo = reinterpret_cast<uint32_t*>(planes.area + (size_t)slot * DP_SLOT_BYTES);
    uint32_t* hi = reinterpret_cast<uint32_t*>(planes.area + (size_t)slot * DP_SLOT_BYTES + DP_PLANE_BYTES);
    // a lane moves four entries per step: two 16-byte loads, one 16-byte store per plane
    for (uint32_t e = e0 + 4u * threadIdx.x; e < e0 + DP_PIECE_ENTRIES && e < entries; e += 4u * DP_THREADS) {
      u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u};
      if (e + 4u <= entries) {
        a = *reinterpret_cast<const u32x4*>(src + 2 * (size_t)e);
        b = *reinterpret_cast<const u32x4*>(src + 2 * (size_t)e + 4);
      } else {  // the dictionary's last one to three entries
        uint32_t t[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        for (uint32_t j = 0; j < 2u * (entries - e); ++j) t[j] = src[2 * (size_t)e + j];
        a = u32x4{t[0], t[1], t[2], t[3]};
        b = u32x4{t[4], t[5], t[6], t[7]};
      }
      const u32x4 l = {a.x, a.z, b.x, b.z}, h = {a.y, a.w, b.y, b.w};
      *reinterpret_cast<u32x4*>(lo + e) = l;  // (whole 16-byte pieces: the planes are padded to them)
      *reinterpret_cast<u32x4*>(hi + e) = h;
    }
  }
}

hipError_t launch_dict_planes(const DictPlanes& planes, const uint32_t* cursor, hipStream_t stream) {
  if (planes.n_slots == 0) return hipSuccess;
  // the number of dictionaries is known to the device only: a fixed grid strides over the work
  const uint32_t grid = planes.n_slots * DP_PIECES < 1024u ? planes.n_slots * DP_PIECES : 1024u;
  hipLaunchKernelGGL(k_dict_planes, dim3(grid), dim3(DP_THREADS), 0, stream, planes, cursor);
  return hipGetLastError();
}

}  // namespace pqd
