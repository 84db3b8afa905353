// DELTA_LENGTH_BYTE_ARRAY and DELTA_BYTE_ARRAY value pages -> offsets + bytes.
//
// The length blocks are already decoded into the page's work area (k_delta_length_blocks).  What is
// left of DeltaLengthByteArrayValuesReader.readBytes (parquet-column/.../values/deltalengthbytearray/
// DeltaLengthByteArrayValuesReader.java:50-58) is a prefix sum of the lengths and one copy of the value
// bytes, which lie back to back behind the length block: one 256-thread workgroup per page.
// DeltaByteArrayReader.readBytes (.../deltastrings/DeltaByteArrayReader.java:63-85) builds value i from
// the first prefix[i] bytes of value i-1 and suffix i, a chain through the whole page.  One wave per
// page takes 64 values at a time: the suffixes of the batch are staged in LDS; then the lanes are byte
// columns (lane l holds byte l and byte l + 64 of the current value in registers) and the values go by
// one after the other: a column inside the value's prefix keeps its byte, a column inside its suffix
// takes the suffix's byte, and the columns write the value into an LDS tile.  Carrying a prefix over
// costs nothing that way, and no step waits for the one before it (the suffix bytes of four values
// are fetched before they are used).  The tile leaves in aligned 16-byte stores.
// Values of more than 128 bytes, or too long for the tiles, are copied one at a time through global memory.
#include "pqd_device.hpp"
#include "pqd_kernels.hpp"
#include "pqd_lz_common.hpp"

namespace pqd {

constexpr int DL_THREADS = 256;
constexpr int DL_PER_LANE = 16;
constexpr uint32_t DL_TILE = DL_THREADS * DL_PER_LANE;
constexpr uint32_t NONE = 0xFFFFFFFFu;
#ifdef DB_PROF
#define DB_STAMP(k) { const uint64_t now_ = __builtin_readcyclecounter(); prof[k] += now_ - prof_t; prof_t = now_; }
#else
#define DB_STAMP(k)
#endif

struct LengthBlocks {
  int32_t status;   // of the length block(s): not PQD_OK fails the page before its first value
  uint32_t n;       // values that have a length in every block
  uint32_t header;  // bytes of the length block(s)
};

__device__ __forceinline__ LengthBlocks length_blocks_of(const pqd_delta_bytes_desc& b) {
  const pqd_page_result* rec = reinterpret_cast<const pqd_page_result*>(b.work);
  LengthBlocks lb;
  const pqd_page_result r0 = rec[0];
  lb.status = r0.status;
  lb.n = r0.values_written < b.num_values ? r0.values_written : b.num_values;
  lb.header = r0.bytes_consumed;
  if (b.encoding == PQD_ENC_DELTA_BYTE_ARRAY && r0.status == PQD_OK) {
    const pqd_page_result r1 = rec[1];
    lb.status = r1.status;
    lb.n = r1.values_written < lb.n ? r1.values_written : lb.n;
    lb.header += r1.bytes_consumed;
  }
  return lb;
}

__global__ __launch_bounds__(DL_THREADS) void k_delta_length_bytes(const pqd_delta_bytes_desc* __restrict__ descs,
                                                                   pqd_page_result* __restrict__ results, uint32_t n_pages) {
  __shared__ uint64_t wave_totals[DL_THREADS / 64];
  __shared__ uint32_t s_stop;
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_delta_bytes_desc b = descs[page];
  if (b.encoding != PQD_ENC_DELTA_LENGTH_BYTE_ARRAY) return;  // the other kernel's page
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const LengthBlocks lb = length_blocks_of(b);
  if (lb.status != PQD_OK) {
    if (tid == 0) {
      b.offsets[0] = 0;
      pqd_page_result r = {lb.status, 0u, 0u, 0u};
      results[page] = r;
    }
    return;
  }
  const int32_t* lengths = reinterpret_cast<const int32_t*>(b.work + DELTA_BYTES_WORK_HEADER);
  const uint32_t n = lb.n;
  const uint64_t avail = b.data_len - lb.header;
  const uint64_t cap = b.bytes ? (b.bytes_capacity > 0x7fffffffu ? 0x7fffffffull : (uint64_t)b.bytes_capacity) : 0x7fffffffull;
  if (tid == 0) s_stop = NONE;
  __syncthreads();
  uint64_t running = 0;
  uint32_t stop = NONE;
  int32_t status = PQD_OK;
  for (uint32_t t0 = 0; t0 < n && stop == NONE; t0 += DL_TILE) {
    const uint32_t first = t0 + tid * DL_PER_LANE;
    int32_t len[DL_PER_LANE];
    uint64_t lane_total = 0;
    bool negative = false;
#pragma unroll
    for (int j = 0; j < DL_PER_LANE; ++j) {
      len[j] = first + j < n ? lengths[first + j] : 0;
      negative |= len[j] < 0;
      lane_total += len[j] < 0 ? 0u : (uint32_t)len[j];
    }
    const uint64_t inc = wave_scan_inclusive(lane_total);
    if (lane == 63) wave_totals[wave] = inc;
    __syncthreads();
    uint64_t base = running + inc - lane_total, tile_total = 0;
#pragma unroll
    for (uint32_t k = 0; k < DL_THREADS / 64; ++k) {
      const uint64_t wt = wave_totals[k];
      base += k < wave ? wt : 0u;
      tile_total += wt;
    }
    const uint64_t limit = avail < cap ? avail : cap;
    if (negative || base + lane_total > limit) {
      uint64_t at = base;
#pragma unroll
      for (int j = 0; j < DL_PER_LANE; ++j) {
        if (first + j < n && (len[j] < 0 || at + (uint32_t)len[j] > limit)) {
          atomicMin(&s_stop, first + j);
          break;
        }
        at += (uint32_t)len[j];
      }
    }
    __syncthreads();
    stop = s_stop;
    uint64_t at = base;
#pragma unroll
    for (int j = 0; j < DL_PER_LANE; ++j) {
      if (first + j < n && first + j <= stop) b.offsets[first + j] = (int32_t)at;
      if (first + j == stop) {
        // this value fails: a broken length or the end of the page (decode error) before the capacity
        status = (len[j] < 0 || at + (uint32_t)len[j] > avail) ? PQD_DECODE : PQD_BAD_ARG;
        wave_totals[0] = at;
        wave_totals[1] = (uint64_t)(uint32_t)status;
      }
      at += len[j] < 0 ? 0u : (uint32_t)len[j];
    }
    __syncthreads();
    if (stop != NONE) {
      running = wave_totals[0];
      status = (int32_t)wave_totals[1];
    } else {
      running += tile_total;
    }
    __syncthreads();
  }
  const uint32_t written = stop != NONE ? stop : n;
  if (stop == NONE && n < b.num_values) status = PQD_DECODE;  // "no more value to read" from the length block
  // the value bytes (they stay in the page when there is no destination)
  if (b.bytes) {
    const uint8_t* src = b.data + lb.header;
    const uint64_t pieces = running >> 4;
    for (uint64_t i = tid; i < pieces; i += DL_THREADS) {
      u32x4 v;
      __builtin_memcpy(&v, src + i * 16, 16);
      __builtin_memcpy(b.bytes + i * 16, &v, 16);
    }
    for (uint64_t i = (pieces << 4) + tid; i < running; i += DL_THREADS) b.bytes[i] = src[i];
  }
  if (tid == 0) {
    if (stop == NONE) b.offsets[n] = (int32_t)running;
    pqd_page_result r = {status, written, lb.header + (uint32_t)running, (uint32_t)running};
    results[page] = r;
  }
}

// ---- DELTA_BYTE_ARRAY ---------------------------------------------------------------------------------

constexpr uint32_t DB_IN = LZ_CHUNK;  // suffix bytes of a batch at most: they lie in the two chunks of the suffix stream that LDS holds
#ifndef DB_STEP
#define DB_STEP 4u  // values per step of the column loop
#endif
constexpr uint32_t DB_OUT = 8192;   // assembled bytes of a batch
constexpr uint32_t DB_COLUMNS = 128; // byte columns held in the lanes' registers: longer values go one at a time

__device__ __forceinline__ uint32_t shuffle_u32(uint32_t v, uint32_t from_lane) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from_lane << 2), (int)v);
}

__device__ __forceinline__ uint8_t load_written_byte(const PQD_GLOBAL uint8_t* p) {
  // bytes this wave stored a moment ago: read them where the stores went (past the vector L1)
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(64) void k_delta_byte_array(const pqd_delta_bytes_desc* __restrict__ descs,
                                                         pqd_page_result* __restrict__ results, uint32_t n_pages) {
  __shared__ __attribute__((aligned(16))) uint8_t in_stage[LZ_STAGE];
  __shared__ __attribute__((aligned(16))) uint8_t out_tile[DB_OUT + 16 + DB_COLUMNS + 16];  // (lanes write up to 128 bytes behind a value's start)
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_delta_bytes_desc b = descs[page];
  if (b.encoding != PQD_ENC_DELTA_BYTE_ARRAY) return;  // the other kernel's page
  const uint32_t lane = threadIdx.x;
  const LengthBlocks lb = length_blocks_of(b);
  if (lb.status != PQD_OK) {
    if (lane == 0) {
      b.offsets[0] = 0;
      pqd_page_result r = {lb.status, 0u, 0u, 0u};
      results[page] = r;
    }
    return;
  }
  const PQD_GLOBAL int32_t* prefix = (const PQD_GLOBAL int32_t*)(b.work + DELTA_BYTES_WORK_HEADER);
  const PQD_GLOBAL int32_t* suffix = prefix + b.num_values;
  PQD_GLOBAL int32_t* offsets = (PQD_GLOBAL int32_t*)b.offsets;
  const uint32_t n = lb.n;
  const PQD_GLOBAL uint8_t* sdata = (const PQD_GLOBAL uint8_t*)b.data + lb.header;
  const uint64_t savail = b.data_len - lb.header;
  const uint64_t cap = b.bytes_capacity > 0x7fffffffu ? 0x7fffffffull : (uint64_t)b.bytes_capacity;

  // the suffix bytes are streamed through LDS, the chunk behind the ones in use already on its way
  InStage in;
  in.base = (const PQD_GLOBAL uint8_t*)((uintptr_t)sdata & ~(uintptr_t)15);
  const uint32_t skew = (uint32_t)((uintptr_t)sdata & 15u);
  in.end = skew + (uint32_t)savail;
  in.lane = lane;
  in.lds = in_stage;
  in.cur = 0;
  in.start(0);
  // the value before the page
  uint32_t last_len = 0;
  const PQD_GLOBAL uint8_t* last_ptr = nullptr;  // its bytes in global memory
  uint32_t c0 = 0, c1 = 0;                        // its bytes lane and lane + 64
  if (b.prev_bytes && b.prev_offsets && b.prev_count) {
    const int32_t p0 = b.prev_offsets[b.prev_count - 1], p1 = b.prev_offsets[b.prev_count];
    last_len = p1 > p0 ? (uint32_t)(p1 - p0) : 0u;
    last_ptr = (const PQD_GLOBAL uint8_t*)b.prev_bytes + p0;
    if (lane < last_len) c0 = last_ptr[lane];
    if (lane + 64u < last_len) c1 = last_ptr[lane + 64u];
  }

  uint32_t i0 = 0;
  uint64_t run_out = 0, run_in = 0;
  int32_t status = PQD_OK;
  // The lengths of the two batches to come are fetched ahead.  (A wait for a load also waits for every store issued
  // before it: lengths fetched one batch ahead would wait for the batch's own output to reach memory, every batch.)
  int32_t P1 = lane < n ? prefix[lane] : 0, S1 = lane < n ? suffix[lane] : 0;
  int32_t P2 = lane + 64u < n ? prefix[lane + 64u] : 0, S2 = lane + 64u < n ? suffix[lane + 64u] : 0;
  uint32_t at1 = 0, at2 = 64;  // the first value of those batches, if the batches in front of them take 64 values each
#ifdef DB_PROF
  uint64_t prof[6] = {0, 0, 0, 0, 0, 0};
  uint64_t prof_t = __builtin_readcyclecounter();
#endif
  while (i0 < n) {
    DB_STAMP(5)
    const uint32_t idx = i0 + lane;
    const bool valid = idx < n;
    if (at1 != i0) {  // a batch was cut short
      P1 = valid ? prefix[idx] : 0;
      S1 = valid ? suffix[idx] : 0;
    }
    const int32_t P = P1, S = S1;
    const bool negative = P < 0 || S < 0;
    const uint32_t p = negative ? 0u : (uint32_t)P, s = negative ? 0u : (uint32_t)S;
    const uint64_t len = (uint64_t)p + s;
    const uint64_t len_before = shuffle_u32((uint32_t)(len > 0xFFFFFFFFull ? 0xFFFFFFFFull : len), (lane + 63u) & 63u);
    const uint64_t prev_len = lane == 0 ? last_len : len_before;
    const uint64_t so_inc = wave_scan_inclusive((uint64_t)s), oo_inc = wave_scan_inclusive(len);
    const uint64_t so = so_inc - s, oo = oo_inc - len;
    const bool broken = valid && (negative || p > prev_len || run_in + so_inc > savail);
    const bool too_big = valid && run_out + oo_inc > cap;
    const uint64_t fail_mask = __ballot(broken || too_big || !valid);
    const uint32_t deliverable = fail_mask ? (uint32_t)__builtin_ctzll(fail_mask) : 64u;
    if (deliverable == 0) {  // lane 0's value fails (it is valid: i0 < n)
      status = __builtin_amdgcn_readfirstlane((int)broken) ? PQD_DECODE : PQD_BAD_ARG;
      break;
    }
    // how many of them fit the tiles
    const bool fits = oo_inc <= DB_OUT && so_inc <= DB_IN && len <= DB_COLUMNS;
    const uint64_t unfit_mask = __ballot(!fits);
    uint32_t m = unfit_mask ? (uint32_t)__builtin_ctzll(unfit_mask) : 64u;
    m = m < deliverable ? m : deliverable;
    PQD_GLOBAL uint8_t* dst = (PQD_GLOBAL uint8_t*)b.bytes + run_out;
    DB_STAMP(0)
    {
      const uint32_t next = i0 + (m ? m : 1u);
      if (at2 == next) {
        P1 = P2;
        S1 = S2;
        at1 = at2;
      } else {
        P1 = next + lane < n ? prefix[next + lane] : 0;
        S1 = next + lane < n ? suffix[next + lane] : 0;
        at1 = next;
      }
      at2 = next + 64u;
      P2 = at2 + lane < n ? prefix[at2 + lane] : 0;
      S2 = at2 + lane < n ? suffix[at2 + lane] : 0;
    }
    if (m == 0) {
      // one value through global memory: its prefix from where the value before it was written
      const uint32_t p0 = __builtin_amdgcn_readfirstlane(p), s0 = __builtin_amdgcn_readfirstlane(s);
      __threadfence();
      for (uint32_t t = lane; t < p0; t += 64) dst[t] = load_written_byte(last_ptr + t);
      const PQD_GLOBAL uint8_t* src = sdata + run_in;
      for (uint32_t t = lane; t < s0; t += 64) dst[p0 + t] = src[t];
      if (lane == 0) offsets[i0] = (int32_t)run_out;
      last_ptr = dst;
      last_len = p0 + s0;
      // the columns' registers: the value's first 128 bytes, read back
      __threadfence();
      if (lane < last_len) c0 = load_written_byte(dst + lane);
      if (lane + 64u < last_len) c1 = load_written_byte(dst + lane + 64u);
      run_out += (uint64_t)p0 + s0;
      run_in += s0;
      i0 += 1;
      continue;
    }
    const uint32_t in_bytes = shuffle_u32((uint32_t)so_inc, m - 1);
    const uint32_t out_bytes = shuffle_u32((uint32_t)oo_inc, m - 1);
    const bool mine = lane < m;
    // the batch's suffix bytes: stream positions first .. first + in_bytes, at most DB_IN of them
    const uint32_t first = skew + (uint32_t)run_in;
    DB_STAMP(1)
    in.reach(first);
    DB_STAMP(2)
    // the tile is laid out like the destination: byte x of the batch at out_tile[a + x]
    const uint32_t a = (uint32_t)((uintptr_t)dst & 15u);
    const uint32_t o = a + (uint32_t)oo;  // this lane's value in the tile
    __syncthreads();  // (one wave: only an order for the LDS accesses)
    // the values in order, a lane per byte column.  Every lane reads and writes for every value: a read outside the
    // value's suffix is not used, a write behind the value's end lands where a later value (or nobody) writes.
    // (prefix and suffix lengths are below 2^16 here, tile positions below 2^14: two values per register to hand round)
    // (bit 12 of the second: the value is longer than 64 bytes.  Lanes behind the batch stand for an empty value at the
    // last value's place: taking them rewrites that value's bytes, which changes nothing)
    const uint32_t o_last = (uint32_t)__builtin_amdgcn_readlane((int)o, (int)(m - 1u));
    const uint32_t ps = mine ? p | (s << 16) : 0u;
    const uint32_t so_o = mine ? ((first + (uint32_t)so - p) & (LZ_STAGE - 1u)) | (p + s > 64u ? 0x1000u : 0u) | (o << 16) : o_last << 16;
    for (uint32_t k0 = 0; k0 < m; k0 += DB_STEP) {
      uint32_t pk[DB_STEP], sk[DB_STEP], to[DB_STEP], base[DB_STEP], r0[DB_STEP], r1[DB_STEP];
      uint32_t wide_bits = 0;
#pragma unroll
      for (uint32_t u = 0; u < DB_STEP; ++u) {
        const uint32_t a_ = (uint32_t)__builtin_amdgcn_readlane((int)ps, (int)(k0 + u)), b_ = (uint32_t)__builtin_amdgcn_readlane((int)so_o, (int)(k0 + u));
        pk[u] = a_ & 0xFFFFu;
        sk[u] = a_ >> 16;
        base[u] = b_ & 0xFFFu;  // where the byte of column 0 would lie in the stage, if the suffix began there
        to[u] = b_ >> 16;
        wide_bits |= b_;
        r0[u] = in_stage[(base[u] + lane) & (LZ_STAGE - 1u)];
      }
      const bool wide = (wide_bits & 0x1000u) != 0u;
      if (wide) {
#pragma unroll
        for (uint32_t u = 0; u < DB_STEP; ++u) r1[u] = in_stage[(base[u] + lane + 64u) & (LZ_STAGE - 1u)];
#pragma unroll
        for (uint32_t u = 0; u < DB_STEP; ++u) {
          c0 = lane - pk[u] < sk[u] ? r0[u] : c0;
          c1 = lane + 64u - pk[u] < sk[u] ? r1[u] : c1;
          out_tile[to[u] + lane] = (uint8_t)c0;
          out_tile[to[u] + lane + 64u] = (uint8_t)c1;
        }
      } else {
#pragma unroll
        for (uint32_t u = 0; u < DB_STEP; ++u) {
          c0 = lane - pk[u] < sk[u] ? r0[u] : c0;
          out_tile[to[u] + lane] = (uint8_t)c0;
        }
      }
    }
    DB_STAMP(3)
    if (mine) offsets[idx] = (int32_t)(run_out + oo);
    __syncthreads();
    // tile -> memory: whole aligned 16-byte pieces, the ragged ends byte by byte
    {
      PQD_GLOBAL uint8_t* gbase = dst - a;  // 16-byte aligned
      const uint32_t end = a + out_bytes;
      for (uint32_t t = lane * 16u; t < end; t += 1024u)
        if (t >= a && t + 16u <= end) *(PQD_GLOBAL u32x4*)(gbase + t) = *reinterpret_cast<const u32x4*>(out_tile + t);
      // the two pieces that are not wholly the batch's: a lane per byte (lanes 0..15 the first piece, 16..31 the last)
      {
        const uint32_t first_piece = a >> 4, last_piece = end >> 4;  // (a < 16: first_piece is 0)
        const bool head = (a & 15u) != 0u, tail = (end & 15u) != 0u && !(head && last_piece == first_piece);
        const uint32_t k = lane < 16u ? first_piece * 16u + lane : last_piece * 16u + (lane - 16u);
        const bool wanted = lane < 16u ? head : lane < 32u && tail;
        if (wanted && k >= a && k < end) gbase[k] = out_tile[k];
      }
    }
    last_len = shuffle_u32((uint32_t)len, m - 1);
    last_ptr = dst + shuffle_u32((uint32_t)oo, m - 1);
    run_out += out_bytes;
    run_in += in_bytes;
    i0 += m;
    __syncthreads();
    DB_STAMP(4)
  }
#ifdef DB_PROF
  if (page == 0 && lane == 0)
    printf("cycles/1000: scans %llu, next lengths %llu, reach %llu, columns %llu, store %llu, loop %llu\n", prof[0] / 1000, prof[1] / 1000,
           prof[2] / 1000, prof[3] / 1000, prof[4] / 1000, prof[5] / 1000);
#endif
  if (status == PQD_OK && i0 < b.num_values) status = PQD_DECODE;  // "no more value to read" from a length block
  if (lane == 0) {
    offsets[i0] = (int32_t)run_out;
    pqd_page_result r = {status, i0, lb.header + (uint32_t)run_in, (uint32_t)run_out};
    results[page] = r;
  }
}

hipError_t launch_delta_bytes(const pqd_delta_bytes_desc* descs, pqd_page_result* results, uint32_t n_pages, hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  hipLaunchKernelGGL(k_delta_length_bytes, dim3(n_pages), dim3(DL_THREADS), 0, stream, descs, results, n_pages);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_delta_byte_array, dim3(n_pages), dim3(64), 0, stream, descs, results, n_pages);
  return hipGetLastError();
}

}  // namespace pqd
