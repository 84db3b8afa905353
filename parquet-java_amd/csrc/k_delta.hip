// DELTA_BINARY_PACKED page decode -> int32 / int64.
//
// Replaces DeltaBinaryPackingValuesReader.initFromPage / loadNewBlockToBuffer / unpackMiniBlock
// (parquet-column/.../values/delta/DeltaBinaryPackingValuesReader.java:60-172) and
// DeltaBinaryPackingConfig.readConfig (DeltaBinaryPackingConfig.java:35-45).  The reference decodes
// the whole page eagerly inside initFromPage; so does this kernel.
//
// One 256-thread workgroup per page:
//   1. page bytes are staged into LDS window by window (coalesced 16-byte loads);
//   2. lane 0 walks the block headers in the window (zig-zag min delta, one width byte per
//      miniblock) and writes one table entry per miniblock {LDS byte offset, width, min delta};
//   3. all lanes unpack: each lane owns four consecutive deltas, sums them with the block's min
//      delta, and the page-wide running sum is formed with a wavefront scan (lane shuffles), a
//      four-entry cross-wave table in LDS and a carry kept in registers from tile to tile.
// Arithmetic wraps modulo 2^64 (2^32 for int32 output, which equals the reference's (int) cast of
// the 64-bit sum).  HBM-bound: w/8 + 14/128 bytes in, 4 or 8 bytes out per value.
#include <type_traits>

#include "pqd_device.hpp"
#include "pqd_kernels.hpp"

namespace pqd {

constexpr int DELTA_THREADS = 256;
constexpr int DWINDOW_BYTES = 18 * 1024;
constexpr int DWINDOW_SLACK = 48;
constexpr int DWINDOW_PIECES = (DWINDOW_BYTES + DWINDOW_SLACK) / 16;
constexpr int MAX_MINIS = 896;      // miniblock table entries per window
constexpr int MAX_MINI_PER_BLOCK = 256;
constexpr int MAX_FAST_BLOCKS = DELTA_THREADS;  // blocks per window whose headers are expanded one per lane

typedef const PQD_LDS uint32_t* lds_words_t;

__device__ __forceinline__ uint32_t ldsb(lds_words_t w, uint32_t off) { return (w[off >> 2] >> ((off & 3u) * 8u)) & 0xFFu; }

struct DeltaState {
  uint32_t pos;           // next unread stream byte
  uint32_t deltas_done;   // deltas expanded before the current window
  uint32_t window_minis;  // table entries of the current window
  uint32_t minis_left;    // miniblocks of the current block still to be emitted
  uint32_t cur_mini;      // index of the next miniblock inside the current block
  int32_t status;
  uint32_t done;
  int32_t window_base;
  uint32_t block_size, mini_n, mini_size, total;
  uint64_t first;
  uint64_t min_delta;     // of the current block
  uint32_t fast_blocks;   // whole blocks found by the quick walk in the current window ...
  uint32_t fast_take;     // ... and the miniblocks of the last of them (the others have mini_n)
  uint32_t fast_base;     // table index of their first miniblock
  uint32_t spec_blocks;   // whole blocks found by wave 0 at the start of the current window ...
  uint32_t spec_bytes;    // ... and the stream bytes they take
  uint32_t spec_len;      // length of the latest block wave 0 found: the first guess in the next window
};

__device__ __forceinline__ bool lds_varint(lds_words_t win, int32_t base, uint32_t& p, uint32_t end,
                                           uint64_t& out, bool is_long) {
  uint64_t v = 0;
  uint32_t shift = 0;
  for (int i = 0; i < 12; ++i) {
    if (p >= end) return false;
    uint64_t b = ldsb(win, p - base);
    ++p;
    uint32_t s = is_long ? (shift & 63u) : (shift & 31u);
    if (b & 0x80u) {
      v |= is_long ? ((b & 0x7Fu) << s) : (uint64_t)(uint32_t)((uint32_t)(b & 0x7Fu) << s);
      shift += 7;
    } else {
      v |= is_long ? (b << s) : (uint64_t)(uint32_t)((uint32_t)b << s);
      out = v;
      return true;
    }
  }
  return false;
}

__device__ __forceinline__ int64_t zigzag64(uint64_t raw) {  // BytesUtils.java:254-258
  return (int64_t)((raw >> 1) ^ (~(raw & 1ull) + 1ull));
}

// Reads the 16 bytes at window offset `off`, where a block header starts: the length of the
// min-delta varint from its continuation bits (`nbv`, 1..8) and the sum of the first `take` width
// bytes behind it (`wsum`).  False if the varint is longer than 8 bytes or a width exceeds 64.
// UNIFORM: the caller is a single lane (or all lanes with the same `off`); the dwords are moved to
// scalar registers so that the bit work runs on the scalar unit, whose dependent operations issue
// back to back (the serial block walk is a chain of these).
template <bool UNIFORM = false>
__device__ __forceinline__ bool quick_block(lds_words_t win, uint32_t off, uint32_t mn, uint32_t take, uint32_t& nbv,
                                            uint32_t& wsum) {
  const uint32_t i0 = off >> 2, sh = (off & 3u) * 8u;
  uint32_t a0 = win[i0], a1 = win[i0 + 1], a2 = win[i0 + 2], a3 = win[i0 + 3], a4 = win[i0 + 4];
  if (UNIFORM) {
    a0 = __builtin_amdgcn_readfirstlane(a0);
    a1 = __builtin_amdgcn_readfirstlane(a1);
    a2 = __builtin_amdgcn_readfirstlane(a2);
    a3 = __builtin_amdgcn_readfirstlane(a3);
    a4 = __builtin_amdgcn_readfirstlane(a4);
  }
  const uint32_t b0 = funnel_shr(a0, a1, sh), b1 = funnel_shr(a1, a2, sh), b2 = funnel_shr(a2, a3, sh), b3 = funnel_shr(a3, a4, sh);
  const uint64_t lo = ((uint64_t)b1 << 32) | b0, hi = ((uint64_t)b3 << 32) | b2;
  const uint64_t stop = ~lo & 0x8080808080808080ull;  // bytes without the continuation bit
  nbv = stop ? ((uint32_t)__builtin_ctzll(stop) >> 3) + 1 : 8u;
  const uint64_t wbytes = nbv == 8 ? hi : ((lo >> (8 * nbv)) | (hi << (64 - 8 * nbv)));
  const uint64_t used = take >= 8 ? wbytes : (wbytes & ((1ull << (8 * take)) - 1ull));
  wsum = __builtin_amdgcn_sad_u8((uint32_t)used, 0u, __builtin_amdgcn_sad_u8((uint32_t)(used >> 32), 0u, 0u));
  (void)mn;
  return stop != 0 && !(used & 0x8080808080808080ull) && !((used + 0x3F3F3F3F3F3F3F3Full) & 0x8080808080808080ull);
}

// One page by one workgroup of DELTA_THREADS threads; `result` gets the page's record (lane 0 writes it).
// All threads of the workgroup call it together; a second call needs a __syncthreads() in between.
template <int ELEM>
__device__ __forceinline__ void delta_page(const pqd_page_desc& d, pqd_page_result* result) {
  typedef typename std::conditional<ELEM == 8, uint64_t, uint32_t>::type acc_t;
  __shared__ u32x4 win4[DWINDOW_PIECES];
  __shared__ uint32_t mini_src[MAX_MINIS];
  __shared__ uint8_t mini_w[MAX_MINIS];  // widths are at most 64; as bytes the kernel fits four workgroups per CU
  __shared__ uint64_t mini_min[MAX_MINIS];
  __shared__ uint8_t cur_widths[MAX_MINI_PER_BLOCK];
  __shared__ acc_t wave_sum[DELTA_THREADS / 64];
  __shared__ uint32_t blk_off[MAX_FAST_BLOCKS];  // window offset of the block headers found by the quick walk
  __shared__ DeltaState st;
  lds_words_t win = (lds_words_t)(const PQD_LDS u32x4*)win4;

  const uint32_t tid = threadIdx.x;
  const uint32_t end = d.data_len;
  const uint8_t* gend = d.data + end;

  if (d.encoding != PQD_ENC_DELTA_BINARY_PACKED || d.elem_size != ELEM) {
    if (tid == 0) {
      pqd_page_result r = {PQD_BAD_ARG, 0u, 0u, 0u};
      *result = r;
    }
    return;
  }

  // ---- page header: stage the first window, lane 0 parses it
  {
    const uint8_t* gbase = reinterpret_cast<const uint8_t*>((uintptr_t)d.data & ~(uintptr_t)15);
    stage_window<DELTA_THREADS>(win4, gbase, gend, DWINDOW_PIECES);
    __syncthreads();
    if (tid == 0) {
      const int32_t base = -(int32_t)(d.data - gbase);
      uint32_t p = 0;
      uint64_t block = 0, mini_n = 0, total = 0, first_raw = 0;
      int32_t status = PQD_OK;
      if (!lds_varint(win, base, p, end, block, false) || !lds_varint(win, base, p, end, mini_n, false)) {
        status = PQD_EOF;
      } else if ((int32_t)block < 0 || (int32_t)mini_n < 0) {
        status = PQD_DECODE;
      } else if (mini_n == 0 || block % mini_n != 0 || ((block / mini_n) & 7u) != 0) {
        status = PQD_BAD_ARG;  // "miniBlockSize must be multiple of 8"
      } else if (!lds_varint(win, base, p, end, total, false)) {
        status = PQD_EOF;
      } else if ((int32_t)total < 0 || total > d.num_values) {
        status = PQD_DECODE;
      } else if (!lds_varint(win, base, p, end, first_raw, true)) {
        status = PQD_EOF;
      } else if (total > 1 && (block == 0 || mini_n > MAX_MINI_PER_BLOCK)) {
        status = PQD_DECODE;
      }
      st.pos = p;
      st.deltas_done = 0;
      st.window_minis = 0;
      st.minis_left = 0;
      st.cur_mini = 0;
      st.status = status;
      st.done = (status != PQD_OK || total <= 1) ? 1u : 0u;
      st.window_base = base;
      st.block_size = (uint32_t)block;
      st.mini_n = (uint32_t)mini_n;
      st.mini_size = mini_n ? (uint32_t)(block / mini_n) : 0u;
      st.total = (uint32_t)total;
      st.first = (uint64_t)zigzag64(first_raw);
      st.min_delta = 0;
      st.spec_len = 0;
    }
    __syncthreads();
  }

  const uint32_t total = st.total;
  const uint32_t mini_size = st.mini_size;
  const uint32_t n_deltas = total > 0 ? total - 1 : 0;
  // miniblocks of 2^k values (the writers' 32): a shift instead of a division per lane and tile
  const uint32_t mini_shift = (mini_size & (mini_size - 1)) == 0 && mini_size ? (uint32_t)__builtin_ctz(mini_size) : ~0u;
  acc_t carry = (acc_t)st.first;  // value at output index deltas_done
  PQD_GLOBAL acc_t* out = (PQD_GLOBAL acc_t*)d.out;
  bool first_window = true;
  constexpr int NEXT_PIECES = (DWINDOW_PIECES + DELTA_THREADS - 1) / DELTA_THREADS;
  u32x4 next_win[NEXT_PIECES];  // the next window on its way from memory while this one is unpacked

  while (st.done == 0) {
    // ---- 1. stage (the header window is reused for the first pass)
    const uint32_t cur = st.pos;
    const uint8_t* gptr = d.data + cur;
    const uint8_t* gbase = reinterpret_cast<const uint8_t*>((uintptr_t)gptr & ~(uintptr_t)15);
    const int32_t wbase = (int32_t)cur - (int32_t)(gptr - gbase);
    if (!first_window) {
      // the window was requested (into registers) before the previous one was unpacked
      __syncthreads();
#pragma unroll
      for (int j = 0; j < NEXT_PIECES; ++j) {
        const int i = (int)tid + j * DELTA_THREADS;
        if (i < DWINDOW_PIECES) win4[i] = keep_leading_bytes(next_win[j], (long)(gend - (gbase + (size_t)i * 16)));
      }
      __syncthreads();
    }
    // ---- 2. block-header walk (loadNewBlockToBuffer, :121-143; readBitWidthsForMiniBlocks, :164-172)
    // 2a. Wave 0, all lanes at once, for stretches of blocks of one length L (steady delta widths):
    // lane t looks at offset t * L behind the stretch's first block, checks that a whole block
    // starts there and works out its length.  The blocks up to the first lane whose length is not L
    // are found, and so is that lane's block (it does start where the lane looked); its length is
    // the guess for the next stretch, so a step costs one look at the window and always advances
    // by a block at least.  The walk ends where a block is not wholly wanted, does not fit the
    // window or does not pass the quick checks: lane 0 carries on from there below.
    uint32_t spec_m = 0, spec_bytes = 0;
    if (tid < 64) {
      const int32_t base = first_window ? st.window_base : wbase;
      const uint32_t wend = (uint32_t)(base + DWINDOW_BYTES);
      const uint32_t mn = st.mini_n, bs = st.block_size;
      uint32_t p = st.pos, scanned = st.deltas_done, found = 0;
      uint32_t L = st.spec_len;  // the latest block length seen in this page (0: none yet)
      // (miniblocks of at most 8192 values: the lengths below stay far inside 32 bits)
      const bool can = st.minis_left == 0 && mn <= 8 && mn > 0 && mini_size <= 8192u;
      while (can && found < (uint32_t)MAX_FAST_BLOCKS) {
        const uint64_t pt = (uint64_t)p + (uint64_t)tid * L;
        const uint32_t off = (uint32_t)(pt - base);
        bool valid = false;
        uint32_t len = 0;
        if (pt + 16 <= end && pt + 16 <= wend && (uint64_t)scanned + (uint64_t)(tid + 1) * bs <= n_deltas &&
            (found + tid + 1) * mn <= (uint32_t)MAX_MINIS && found + tid < (uint32_t)MAX_FAST_BLOCKS) {
          uint32_t nbv, wsum;
          valid = quick_block(win, off, mn, mn, nbv, wsum);
          len = nbv + mn + (mini_size >> 3) * wsum;
          valid = valid && pt + len <= end && pt + len <= wend;
        }
        const uint64_t no = ~__builtin_amdgcn_ballot_w64(valid && len == L);
        const uint32_t m = no ? (uint32_t)__builtin_ctzll(no) : 64u;  // lanes 0..m-1: blocks of length L
        // lane m looked at the right place too (every block before it has length L)
        const uint32_t lm = m < 64u ? m : 0u;
        const bool valid_m = m < 64u && __builtin_amdgcn_readlane((uint32_t)valid, lm) != 0u;
        const uint32_t len_m = __builtin_amdgcn_readlane(len, lm);
        const uint32_t take = m + (valid_m ? 1u : 0u);
        if (take == 0) break;
        if (tid < take) blk_off[found + tid] = off;
        found += take;
        p += m * L + (valid_m ? len_m : 0u);
        scanned += take * bs;
        if (m < 64u) {
          if (!valid_m) break;
          L = len_m;
        }
      }
      if (tid == 0) st.spec_len = L;
      if (tid == 0) {
        st.spec_blocks = found;
        st.spec_bytes = p - st.pos;
      }
    }
    __syncthreads();
    spec_m = st.spec_blocks;
    spec_bytes = st.spec_bytes;
    if (tid == 0) {
      const int32_t base = first_window ? st.window_base : wbase;
      st.window_base = base;
      const uint32_t wend = (uint32_t)(base + DWINDOW_BYTES);
      uint32_t p = st.pos, nm = 0, left = st.minis_left, cm = st.cur_mini;
      uint32_t scanned = st.deltas_done;
      uint64_t min_delta = st.min_delta;
      int32_t status = PQD_OK;
      const uint32_t mini_bytes_unit = mini_size >> 3;
      auto careful_walk = [&](bool only_current_block) {
      while (nm < MAX_MINIS && scanned < n_deltas && status == PQD_OK) {
        if (left == 0) {
          if (only_current_block) break;
          if (nm > 0 && wend < end && p + 12 + st.mini_n > wend) break;  // header may straddle: restage
          uint64_t raw;
          if (!lds_varint(win, base, p, end, raw, true)) {
            status = PQD_DECODE;  // "can not read min delta in current block"
            break;
          }
          min_delta = (uint64_t)zigzag64(raw);
          if (end - p < st.mini_n) {
            status = PQD_DECODE;  // "Can not decode bitwidth in block header"
            break;
          }
          for (uint32_t i = 0; i < st.mini_n; ++i) cur_widths[i] = (uint8_t)ldsb(win, p - base + i);
          p += st.mini_n;
          const uint32_t need = (n_deltas - scanned + mini_size - 1) / mini_size;
          left = need < st.mini_n ? need : st.mini_n;
          cm = 0;
        }
        const uint32_t w = cur_widths[cm];
        if (w > 64) {  // the packer table of the reference has 65 entries
          status = PQD_DECODE;
          break;
        }
        const uint64_t bytes = (uint64_t)mini_bytes_unit * w;  // up to 2^28 * 64: not a 32-bit product
        if (end - p < bytes) {
          status = PQD_EOF;  // in.slice(w) past the end
          break;
        }
        if ((uint64_t)p + bytes > wend) {
          if (nm == 0) status = PQD_DECODE;  // a single miniblock larger than the staging window
          break;
        }
        mini_src[nm] = (uint32_t)(p - base);
        mini_w[nm] = (uint8_t)w;
        mini_min[nm] = min_delta;
        ++nm;
        p += (uint32_t)bytes;
        scanned += mini_size;
        ++cm;
        --left;
      }
      };
      careful_walk(true);  // the rest of a block that the previous window cut
      const uint32_t fast_base = nm;
      // Quick walk over whole blocks (lane 0 only finds where they start; their headers are expanded
      // into the miniblock table by one lane per block below).  It reads 16 bytes per block, takes
      // the length of the min-delta varint from its continuation bits and adds up the width bytes.
      // Anything unusual (more than 8 miniblocks per block, a varint longer than 8 bytes, a width
      // above 64, the end of the stream or of the window in sight) ends it, and the careful walk
      // below carries on from that block and reports what is wrong with it.
      uint32_t nblk = 0, last_take = st.mini_n;
      bool window_full = false;  // the walk stopped in front of a whole block that does not fit the window
      if (left == 0 && st.mini_n <= 8 && mini_size <= 8192u && status == PQD_OK) {
        const uint32_t mn = st.mini_n;
        // the blocks found by all lanes above
        nblk = spec_m;
        nm += spec_m * mn;
        p += spec_bytes;
        scanned += spec_m * st.block_size;
        p = __builtin_amdgcn_readfirstlane(p);
        nm = __builtin_amdgcn_readfirstlane(nm);
        scanned = __builtin_amdgcn_readfirstlane(scanned);
        nblk = __builtin_amdgcn_readfirstlane(nblk);
        // miniblocks still wanted (one division here instead of one per block)
        uint32_t left_minis = __builtin_amdgcn_readfirstlane((n_deltas - scanned + mini_size - 1) / mini_size);
        while (nm + mn <= MAX_MINIS && nblk < (uint32_t)MAX_FAST_BLOCKS && scanned < n_deltas) {
          if (p + 16 > end) break;
          if (p + 16 > wend) {
            window_full = nm > 0;
            break;
          }
          const uint32_t off = (uint32_t)(p - base);
          const uint32_t take = left_minis < mn ? left_minis : mn;
          uint32_t nbv, wsum;
          if (!quick_block<true>(win, off, mn, take, nbv, wsum)) break;
          const uint32_t pay = p + nbv + mn, bytes = mini_bytes_unit * wsum;
          if (end - pay < bytes) break;
          if (pay + bytes > wend) {
            window_full = nm > 0;
            break;
          }
          blk_off[nblk] = off;
          ++nblk;
          last_take = take;
          nm += take;
          p = pay + bytes;
          scanned += take * mini_size;
          left_minis -= take;
        }
      }
      st.fast_blocks = nblk;
      st.fast_take = last_take;
      st.fast_base = fast_base;
      if (!window_full) careful_walk(false);  // (a window that ends between blocks lets the next one start with all lanes)
      st.pos = p;
      st.window_minis = nm;
      st.minis_left = left;
      st.cur_mini = cm;
      st.min_delta = min_delta;
      if (status != PQD_OK) {
        st.status = status;
        st.done = 1;
      } else if (scanned >= n_deltas) {
        st.done = 1;
      }
    }
    __syncthreads();
    first_window = false;
    if (st.done == 0 && st.status == PQD_OK) {  // request the next window now; it lands behind the unpacking
      const uint8_t* nptr = d.data + st.pos;
      const uint8_t* nbase = reinterpret_cast<const uint8_t*>((uintptr_t)nptr & ~(uintptr_t)15);
#pragma unroll
      for (int j = 0; j < NEXT_PIECES; ++j) {
        const int i = (int)tid + j * DELTA_THREADS;
        const uint8_t* q = nbase + (size_t)i * 16;
        next_win[j] = u32x4{0u, 0u, 0u, 0u};
        if (i < DWINDOW_PIECES && q < gend) next_win[j] = *(const PQD_GLOBAL u32x4*)(q);
      }
    }
    // one lane per block found by the quick walk: min delta, widths -> miniblock table entries
    {
      const uint32_t nblk = st.fast_blocks, mn = st.mini_n;
      if (tid < nblk) {
        const uint32_t off = blk_off[tid];
        uint32_t q = off;
        uint64_t raw = 0;
        uint32_t shift = 0;
        for (int i = 0; i < 8; ++i) {  // the quick walk made sure the varint ends within 8 bytes
          const uint64_t b = ldsb(win, q);
          ++q;
          raw |= (b & 0x7Fu) << shift;
          shift += 7;
          if (!(b & 0x80u)) break;
        }
        const uint64_t md = (uint64_t)zigzag64(raw);
        const uint32_t take = tid + 1 == nblk ? st.fast_take : mn;
        uint32_t src = q + mn;
        for (uint32_t i = 0; i < take; ++i) {
          const uint32_t w = ldsb(win, q + i);
          const uint32_t e = st.fast_base + tid * mn + i;
          mini_src[e] = src;
          mini_w[e] = (uint8_t)w;
          mini_min[e] = md;
          src += (mini_size >> 3) * w;
        }
      }
      __syncthreads();
    }
    if (st.status != PQD_OK) break;

    // ---- 3. unpack + running sum, 1024 deltas per tile
    const uint32_t D0 = st.deltas_done;
    const uint32_t wd = st.window_minis * mini_size;
    for (uint32_t t0 = 0; t0 < wd; t0 += DELTA_THREADS * 4) {
      const uint32_t l = t0 + tid * 4;
      acc_t dv[4] = {0, 0, 0, 0};
      acc_t mn = 0;
      if (l < wd) {
        const uint32_t m = mini_shift != ~0u ? l >> mini_shift : l / mini_size;
        const uint32_t w = mini_w[m];
        mn = (acc_t)mini_min[m];
        uint32_t bit = mini_src[m] * 8 + (l - m * mini_size) * w;
        if (ELEM == 8) {
          const uint64_t mask = mask64(w);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t i = bit >> 5, sh = bit & 31u;
            const uint32_t a = win[i], b = win[i + 1], c = win[i + 2];
            const uint64_t v = ((uint64_t)funnel_shr(b, c, sh) << 32) | funnel_shr(a, b, sh);
            dv[j] = (acc_t)(v & mask);
            bit += w;
          }
        } else {
          const uint32_t mask = mask32(w > 32 ? 32 : w);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t i = bit >> 5, sh = bit & 31u;
            dv[j] = (acc_t)(funnel_shr(win[i], win[i + 1], sh) & mask);
            bit += w;
          }
        }
      }
      const acc_t s = (acc_t)(mn * 4 + dv[0] + dv[1] + dv[2] + dv[3]);
      // inclusive scan over the wave
      const acc_t inc = wave_scan_inclusive(s);
      const uint32_t lane = tid & 63u;
      if (lane == 63) wave_sum[tid >> 6] = inc;
      __syncthreads();
      acc_t before = 0, tile_total = 0;
#pragma unroll
      for (int k = 0; k < DELTA_THREADS / 64; ++k) {
        const acc_t ws = wave_sum[k];
        if ((uint32_t)k < (tid >> 6)) before += ws;
        tile_total += ws;
      }
      const acc_t o0 = (acc_t)(carry + before + inc - s);
      if (l < wd) {
        const acc_t o1 = (acc_t)(o0 + mn + dv[0]);
        const acc_t o2 = (acc_t)(o1 + mn + dv[1]);
        const acc_t o3 = (acc_t)(o2 + mn + dv[2]);
        const uint32_t g0 = D0 + l;
        if (g0 + 4 <= total) {
          if (ELEM == 8) {
            u32x4_a4 r0 = {(uint32_t)o0, (uint32_t)((uint64_t)o0 >> 32), (uint32_t)o1, (uint32_t)((uint64_t)o1 >> 32)};
            u32x4_a4 r1 = {(uint32_t)o2, (uint32_t)((uint64_t)o2 >> 32), (uint32_t)o3, (uint32_t)((uint64_t)o3 >> 32)};
            *(PQD_GLOBAL u32x4_a4*)(out + g0) = r0;
            *(PQD_GLOBAL u32x4_a4*)(out + g0 + 2) = r1;
          } else {
            u32x4_a4 r = {(uint32_t)o0, (uint32_t)o1, (uint32_t)o2, (uint32_t)o3};
            *(PQD_GLOBAL u32x4_a4*)(out + g0) = r;
          }
        } else {
          if (g0 < total) out[g0] = o0;
          if (g0 + 1 < total) out[g0 + 1] = o1;
          if (g0 + 2 < total) out[g0 + 2] = o2;
          if (g0 + 3 < total) out[g0 + 3] = o3;
        }
      }
      carry = (acc_t)(carry + tile_total);
      __syncthreads();  // wave_sum is reused by the next tile
    }
    if (tid == 0) st.deltas_done = D0 + wd;
    __syncthreads();
  }

  if (tid == 0) {
    pqd_page_result r;
    r.status = st.status;
    r.reserved = 0;
    if (st.status == PQD_OK) {
      // the value that follows the last unpacked delta (also the lone header value of a 1-value page)
      if (st.deltas_done < total) out[st.deltas_done] = carry;
      r.values_written = total;
      r.bytes_consumed = st.pos;
    } else {
      r.values_written = 0;
      r.bytes_consumed = 0;
    }
    *result = r;
  }
}

template <int ELEM>
__global__ __launch_bounds__(DELTA_THREADS) void k_delta_pages(const pqd_page_desc* __restrict__ descs,
                                                               pqd_page_result* __restrict__ results,
                                                               uint32_t n_pages) {
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_page_desc d = descs[page];
  delta_page<ELEM>(d, results + page);
}

// The length blocks of DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY pages, decoded into the page's work
// area: DeltaLengthByteArrayValuesReader.initFromPage reads all lengths first (parquet-column/.../values/
// deltalengthbytearray/DeltaLengthByteArrayValuesReader.java:44-48), DeltaByteArrayReader.initFromPage
// reads the prefix lengths and then, from where they end, the suffix lengths (.../deltastrings/
// DeltaByteArrayReader.java:50-54).  Work area: two result records (32 bytes, padded to 64), the first
// block's int32 values, then (DELTA_BYTE_ARRAY) the second block's.
__global__ __launch_bounds__(DELTA_THREADS) void k_delta_length_blocks(const pqd_delta_bytes_desc* __restrict__ descs,
                                                                       uint32_t n_pages) {
  __shared__ pqd_page_result first_block;
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_delta_bytes_desc b = descs[page];
  if (b.encoding == PQD_ENC_PLAIN_BYTE_ARRAY) return;  // no length blocks (and no work area)
  pqd_page_result* rec = reinterpret_cast<pqd_page_result*>(b.work);
  int32_t* lengths = reinterpret_cast<int32_t*>(b.work + DELTA_BYTES_WORK_HEADER);
  pqd_page_desc d{};
  d.data = b.data;
  d.out = lengths;
  d.data_len = b.data_len;
  d.num_values = b.num_values;
  d.encoding = PQD_ENC_DELTA_BINARY_PACKED;
  d.elem_size = 4;
  delta_page<4>(d, &first_block);
  __syncthreads();
  const pqd_page_result r0 = first_block;
  if (threadIdx.x == 0) rec[0] = r0;
  if (b.encoding != PQD_ENC_DELTA_BYTE_ARRAY) return;
  if (r0.status != PQD_OK) {
    if (threadIdx.x == 0) rec[1] = r0;
    return;
  }
  d.data = b.data + r0.bytes_consumed;
  d.data_len = b.data_len - r0.bytes_consumed;
  d.out = lengths + b.num_values;
  __syncthreads();
  delta_page<4>(d, rec + 1);
}

template __global__ void k_delta_pages<4>(const pqd_page_desc*, pqd_page_result*, uint32_t);
template __global__ void k_delta_pages<8>(const pqd_page_desc*, pqd_page_result*, uint32_t);


hipError_t launch_delta_length_blocks(const pqd_delta_bytes_desc* descs, uint32_t n_pages, hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  hipLaunchKernelGGL(k_delta_length_blocks, dim3(n_pages), dim3(DELTA_THREADS), 0, stream, descs, n_pages);
  return hipGetLastError();
}

hipError_t launch_delta_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages, int elem,
                              hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  const dim3 grid(n_pages), block(DELTA_THREADS);
  if (elem == 4)
    hipLaunchKernelGGL((k_delta_pages<4>), grid, block, 0, stream, descs, results, n_pages);
  else
    hipLaunchKernelGGL((k_delta_pages<8>), grid, block, 0, stream, descs, results, n_pages);
  return hipGetLastError();
}

}  // namespace pqd
