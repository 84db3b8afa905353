// PLAIN pages of BYTE_ARRAY values -> offsets + bytes.
//
// Replaces the read loop over BinaryPlainValuesReader.readBytes (parquet-column/.../values/plain/
// BinaryPlainValuesReader.java:36-74: length = the next four bytes, little-endian; the value = the `length` bytes
// behind them; initFromPage slices everything that is available, a value that cannot be read is a
// ParquetDecodingException "could not read bytes at offset").  The page is a chain: where value i + 1 starts is
// known only once the length of value i is read.  One wave per page takes the page in windows of 4 KiB held in
// LDS, one 64-byte segment per lane:
//   1. every lane walks its segment backwards and notes, for each of the 64 places a value could start, where
//      the chain from there leaves the segment (a place's entry is that of the place its own length points at,
//      which lies behind it: one look-up per place);
//   2. the wave follows the chain through the window, one table look-up per segment entered;
//   3. every lane that was entered walks its segment forward from its entry, twice: first it counts its values (a
//      prefix sum over the lanes then says which value of the page each lane starts with), then it lists their
//      starts (offsets[i] = start - 4 * i: everything in front of a value but the length fields is value bytes);
//   4. the value bytes are squeezed together: byte q of the page goes to q - 4 * (starts at or before q), the
//      count taken from the lanes' start masks; the window's bytes and offsets leave through LDS tiles in
//      aligned 16-byte stores.
// Windows that lie inside one long value are copied straight from memory to memory.
// Status: PQD_DECODE at the first value whose length field or bytes do not lie inside the page or whose length is
// negative; PQD_BAD_ARG at the first value that does not fit into bytes_capacity.  bytes_consumed = the page
// bytes in front of the first value not delivered.
#include "pqd_device.hpp"
#include "pqd_kernels.hpp"

namespace pqd {

constexpr uint32_t PB_WIN = 4096;
constexpr uint32_t PB_NONE = 0xFFFFFFFFu;
// a table entry: where the chain from a place leaves its segment, as exit - w0; PB_BEYOND: at or behind the window's
// end; PB_FAIL: the chain stops inside the segment (a value that cannot be read).  16 bits: with the table, the window
// and the tiles a page takes 14 KB of LDS, and what counts for a call is how many pages are in flight at once.
constexpr uint32_t PB_FAIL = 0xFFFFu, PB_BEYOND = 0xFFFEu;

// byte q of the window in LDS: four bytes of padding behind every 64-byte segment, so that the lanes' segments
// start in different banks
__device__ __forceinline__ uint32_t pb_at(uint32_t q) { return q + ((q >> 6) << 2); }

struct PbLds {
  uint8_t win[PB_WIN + 64 + 4 * 66];
  union {
    uint16_t exits[64 * 65];  // steps 1 and 2: per place, where the chain from it leaves the segment
    struct {                  // steps 3 and 4
      uint32_t offs[PB_WIN / 4 + 4];
      __attribute__((aligned(16))) uint8_t out[PB_WIN + 48];
      uint64_t starts[64];   // per segment: places where a value starts
      uint64_t lengths[64];  // per segment: bytes of length fields
      uint32_t before[64];   // values that start in front of the segment (from the page's first)
    } t;
  } u;
};

__device__ __forceinline__ uint32_t pb_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// The four bytes at window place q as a little-endian number.
__device__ __forceinline__ uint32_t pb_length_at(const uint8_t* win, uint32_t q) {
  return (uint32_t)win[pb_at(q)] | ((uint32_t)win[pb_at(q + 1)] << 8) | ((uint32_t)win[pb_at(q + 2)] << 16) |
         ((uint32_t)win[pb_at(q + 3)] << 24);
}

// Can a value that starts at page position `pos` be read (its length is `len`)?  Then `next` = where the value
// behind it starts.
__device__ __forceinline__ bool pb_value_at(uint32_t pos, uint32_t len, uint32_t data_len, uint32_t& next) {
  // (pos and data_len are below 2^31, and so is a length that is not negative: the sums fit 32 bits)
  next = pos + 4u + len;
  return (int32_t)len >= 0 && pos + 4u <= data_len && next <= data_len;
}

__global__ __launch_bounds__(64) void k_plain_byte_array(const pqd_delta_bytes_desc* __restrict__ descs,
                                                         pqd_page_result* __restrict__ results, uint32_t n_pages) {
  __shared__ PbLds L;
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_delta_bytes_desc b = descs[page];
  if (b.encoding != PQD_ENC_PLAIN_BYTE_ARRAY) return;  // the other kernels' page
  const uint32_t lane = threadIdx.x;
  PQD_GLOBAL int32_t* offsets = (PQD_GLOBAL int32_t*)b.offsets;
  if (b.data_len > 0x7fffffffu) {  // a page is a Java buffer
    if (lane == 0) {
      offsets[0] = 0;
      pqd_page_result r = {PQD_BAD_ARG, 0u, 0u, 0u};
      results[page] = r;
    }
    return;
  }
  const PQD_GLOBAL uint8_t* data = (const PQD_GLOBAL uint8_t*)b.data;
  PQD_GLOBAL uint8_t* bytes = (PQD_GLOBAL uint8_t*)b.bytes;
  const uint32_t data_len = b.data_len, n = b.num_values;
  const uint32_t cap = bytes ? (b.bytes_capacity > 0x7fffffffu ? 0x7fffffffu : b.bytes_capacity) : 0x7fffffffu;

#ifdef PB_PROF  // development: cycles of each step, summed over the page, printed by page 0
  uint64_t pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_last = __builtin_readcyclecounter();
#define PB_T(k) { const uint64_t now_ = __builtin_readcyclecounter(); pt[k] += now_ - pt_last; pt_last = now_; }
#else
#define PB_T(k)
#endif
  uint32_t w0 = 0;       // page position of the window
  uint32_t cur = 0;      // where the next value starts
  uint32_t idx = 0;      // values delivered
  uint32_t run_out = 0;  // value bytes written
  uint32_t spill = 0;    // bytes at the window's start that belong to a length field of the window before
  int32_t status = PQD_OK;
  bool stopped = n == 0;
  while (!stopped || w0 < cur) {
    if (stopped || cur >= w0 + PB_WIN) {
      // whole windows inside one value, or what is left of the last value delivered: memory to memory
      const uint32_t span = stopped ? cur - w0 : ((cur - w0) / PB_WIN) * PB_WIN;
      if (bytes) {
        const PQD_GLOBAL uint8_t* src = data + w0 + spill;
        PQD_GLOBAL uint8_t* dst = bytes + run_out;
        const uint32_t count = span - spill, pieces = count >> 4;
        for (uint32_t i = lane; i < pieces; i += 64u) {
          u32x4 v;
          __builtin_memcpy(&v, (const uint8_t*)(src + i * 16u), 16);
          __builtin_memcpy((uint8_t*)(dst + i * 16u), &v, 16);
        }
        for (uint32_t i = (pieces << 4) + lane; i < count; i += 64u) dst[i] = src[i];
      }
      run_out += span - spill;
      spill = 0;
      w0 += span;
      continue;
    }
    // ---- the window and some bytes behind it -> LDS
    {
      const uint32_t want = PB_WIN + 16u, have = data_len - w0 < want ? data_len - w0 : want;
      const PQD_GLOBAL uint8_t* src = data + w0;
      for (uint32_t t = lane * 16u; t < have; t += 1024u) {
        if (t + 16u <= have) {
          u32x4 v;
          __builtin_memcpy(&v, (const uint8_t*)(src + t), 16);
          uint32_t* d = reinterpret_cast<uint32_t*>(L.win + pb_at(t));  // a piece never crosses a segment's end
          d[0] = v.x;
          d[1] = v.y;
          d[2] = v.z;
          d[3] = v.w;
        } else {
          for (uint32_t k = 0; t + k < have; ++k) L.win[pb_at(t + k)] = src[t + k];
        }
      }
    }
    __syncthreads();  // (one wave: an order for the LDS accesses)
    PB_T(0)
    // ---- 1. the segment's places, last to first, four at a time (a value is its four length bytes at least: none of
    // the four continues at another).  The lane's 64 bytes and the four behind them sit in registers: a place's
    // length field is a funnel shift, and the only trips to LDS are the look-ups of the places the values end at.
    const uint32_t seg = w0 + lane * 64u;
    {
      uint32_t sb[17];
      const uint32_t* own = reinterpret_cast<const uint32_t*>(L.win + lane * 68u);  // (pb_at: 64 bytes, four of padding)
#pragma unroll
      for (int k = 0; k < 16; ++k) sb[k] = own[k];
      sb[16] = own[17];
#pragma unroll
      for (int g = 15; g >= 0; --g) {
        uint32_t next[4], j[4], e_j[4];
        bool ok[4], inside[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int p = 4 * g + 3 - u;
          const uint32_t len = (p & 3) ? funnel_shr(sb[p >> 2], sb[(p >> 2) + 1], (uint32_t)(p & 3) * 8u) : sb[p >> 2];
          ok[u] = pb_value_at(seg + (uint32_t)p, len, data_len, next[u]);
          inside[u] = ok[u] && next[u] < seg + 64u;
          j[u] = inside[u] ? next[u] - seg : 63u;  // (the look-up is made in any case)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) e_j[u] = L.u.exits[lane * 65u + j[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int p = 4 * g + 3 - u;
          const uint32_t rel = next[u] - w0 < PB_WIN ? next[u] - w0 : PB_BEYOND;  // (at or beyond the window's end: the chain ends there)
          L.u.exits[lane * 65u + (uint32_t)p] = (uint16_t)(!ok[u] ? PB_FAIL : inside[u] ? e_j[u] : rel);
        }
      }
    }
    __syncthreads();
    PB_T(1)
    // ---- 2. the chain through the window (how many values it passes is counted in step 3)
    const uint32_t idx0 = idx;
    uint32_t entry = PB_NONE;
    {
      uint32_t at = cur;
      while (at < w0 + PB_WIN) {
        const uint32_t k = (at - w0) >> 6, p = (at - w0) & 63u;
        const uint32_t e = pb_uni(L.u.exits[k * 65u + p]);
        if (lane == k) entry = at;
        if (e >= PB_BEYOND) break;
        at = w0 + e;
      }
    }
    __syncthreads();  // the tables are done with: their room becomes the tiles
    PB_T(2)
    // ---- 3. the values of this lane's segment: counted first (a value that cannot be read ends the count; the chain
    // may have run on behind the page's last value: lanes that start at or behind it drop out), then listed
    uint32_t first_index = idx0;
    {
      uint32_t count = 0;
      if (entry != PB_NONE) {
        uint32_t q = entry;
        while (q < seg + 64u) {
          uint32_t next;
          if (!pb_value_at(q, pb_length_at(L.win, q - w0), data_len, next)) break;
          ++count;
          q = next;
        }
      }
      first_index += wave_scan_inclusive(count) - count;
      if (first_index >= n) entry = PB_NONE;
    }
    uint64_t starts = 0;
    uint32_t mine = 0, end = 0;
    int32_t fail = PQD_OK;
    if (entry != PB_NONE) {
      uint32_t q = entry, i = first_index;
      while (q < seg + 64u && i < n) {
        uint32_t next;
        if (!pb_value_at(q, pb_length_at(L.win, q - w0), data_len, next)) {
          fail = PQD_DECODE;
          break;
        }
        if (next - 4u * (i + 1u) > cap) {
          fail = PQD_BAD_ARG;
          break;
        }
        L.u.t.offs[i - idx0] = q - 4u * i;
        starts |= 1ull << (q - seg);
        ++i;
        q = next;
      }
      mine = i - first_index;
      end = q;
    }
    // lanes behind one that stopped were entered for nothing (a value that did not fit)
    {
      const uint64_t stops = __ballot(fail != PQD_OK);
      if (stops && lane > (uint32_t)__builtin_ctzll(stops)) {
        entry = PB_NONE;
        starts = 0;
        mine = 0;
        fail = PQD_OK;
      }
    }
    const uint64_t entered = __ballot(entry != PB_NONE);
    const uint32_t inc = wave_scan_inclusive(mine);
    const uint32_t in_window = pb_uni((uint32_t)__builtin_amdgcn_readlane((int)inc, 63));
    if (entered) {
      const uint32_t last = pb_uni(63u - (uint32_t)__builtin_clzll(entered));
      cur = (uint32_t)__builtin_amdgcn_readlane((int)end, last);
      status = __builtin_amdgcn_readlane(fail, last);
    }
    idx = idx0 + in_window;
    stopped = status != PQD_OK || idx >= n;
    // bytes of the window that are delivered: those in front of `cur`
    const uint32_t limit = cur < w0 + PB_WIN ? cur : w0 + PB_WIN;
    // length fields: four bytes from every start, the last ones may lie in the segment (or window) behind
    uint64_t lens = starts | (starts << 1) | (starts << 2) | (starts << 3);
    const uint32_t over = (uint32_t)(starts >> 61);  // bit 0: three bytes here, one behind; bit 1: two; bit 2: three
    uint32_t behind = (over & 4u) ? 7u : (over & 2u) ? 3u : (over & 1u) ? 1u : 0u;
    uint32_t from_before = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane + 63u) & 63u) << 2), (int)behind);
    if (lane == 0) from_before = spill ? (1u << spill) - 1u : 0u;
    lens |= from_before;
    const uint32_t spill_next = __builtin_popcount(pb_uni((uint32_t)__builtin_amdgcn_readlane((int)behind, 63)));
    L.u.t.starts[lane] = starts;
    L.u.t.lengths[lane] = lens;
    L.u.t.before[lane] = idx0 + inc - mine;
    __syncthreads();
    PB_T(3)
    // ---- 4. the value bytes, squeezed together
    // (when the window ends inside a length field, the bytes of that field behind the window are not in `limit`)
    const uint32_t out_end = limit - 4u * idx + (limit == w0 + PB_WIN ? spill_next : 0u);
    if (bytes) {
      const uint32_t a = (uint32_t)((uintptr_t)(bytes + run_out) & 15u);
      const uint64_t upto = (2ull << lane) - 1ull;
      for (uint32_t t0 = 0; t0 < 64u && w0 + t0 * 64u < limit; t0 += 4u) {  // four segments side by side
        uint64_t s[4], l[4];
        uint32_t bf[4], byte[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
          s[u] = L.u.t.starts[t0 + u];
          l[u] = L.u.t.lengths[t0 + u];
          bf[u] = L.u.t.before[t0 + u];
          byte[u] = L.win[pb_at((t0 + u) * 64u + lane)];
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
          const uint32_t q = w0 + (t0 + u) * 64u + lane;
          if (q < limit && !((l[u] >> lane) & 1ull)) {
            const uint32_t dst = q - 4u * (bf[u] + (uint32_t)__builtin_popcountll(s[u] & upto));
            L.u.t.out[a + (dst - run_out)] = (uint8_t)byte[u];
          }
        }
      }
      __syncthreads();
      PB_T(4)
      PQD_GLOBAL uint8_t* gbase = bytes + run_out - a;  // 16-byte aligned
      const uint32_t stop_at = a + (out_end - run_out);
      for (uint32_t t = lane * 16u; t < stop_at; t += 1024u) {
        if (t >= a && t + 16u <= stop_at) {
          *(PQD_GLOBAL u32x4*)(gbase + t) = *reinterpret_cast<const u32x4*>(L.u.t.out + t);
        } else {
          for (uint32_t k = 0; k < 16u; ++k)
            if (t + k >= a && t + k < stop_at) gbase[t + k] = L.u.t.out[t + k];
        }
      }
    }
    for (uint32_t t = lane; t < in_window; t += 64u) offsets[idx0 + t] = (int32_t)L.u.t.offs[t];
    run_out = out_end;
    spill = spill_next;
    w0 += PB_WIN;
    __syncthreads();
    PB_T(5)
  }
#ifdef PB_PROF
  if (page == 0 && lane == 0)
    printf("k_plain_byte_array, page 0, cycles/1000: stage %llu, places %llu, chain %llu, values %llu, squeeze %llu, stores %llu\n", pt[0] / 1000,
           pt[1] / 1000, pt[2] / 1000, pt[3] / 1000, pt[4] / 1000, pt[5] / 1000);
#endif
  if (lane == 0) {
    offsets[idx] = (int32_t)run_out;
    pqd_page_result r = {status, idx, cur, run_out};
    results[page] = r;
  }
}

hipError_t launch_plain_byte_array(const pqd_delta_bytes_desc* descs, pqd_page_result* results, uint32_t n_pages, hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  hipLaunchKernelGGL(k_plain_byte_array, dim3(n_pages), dim3(64), 0, stream, descs, results, n_pages);
  return hipGetLastError();
}

}  // namespace pqd
