// Header walk of RLE / bit-packing hybrid pages: one wave per page, thousands of pages in flight.
//
// Follows RunLengthBitPackingHybridDecoder.readNext (parquet-column/.../values/rle/
// RunLengthBitPackingHybridDecoder.java:80-109) run by run, but instead of unpacking it records
// where each run's values come from (pqd_rle_plan.hpp).  Stream prefixes are handled here too:
// the 4-byte length of RunLengthBitPackingHybridValuesReader (:41-45) and the 1-byte width of
// DictionaryValuesReader (parquet-column/.../values/dictionary/DictionaryValuesReader.java:49-63).
//
// Only header bytes are touched (a few bytes per run); the payload is never read here.  The wave
// fetches headers speculatively: each of its 64 lanes loads 16 bytes at pos + lane*stride, where
// stride is the length of the run just walked (16 while nothing is known, which makes the 64
// blocks one contiguous KiB).  Pages written by parquet-java consist of equal-length 504-value
// runs, so one such round brings in every remaining header of the page; irregular pages simply
// take more rounds.  Latency-bound by design: the chip hides it by running one wave per page for
// every page of the batch at once.
#include <cstdlib>

#include "pqd_device.hpp"
#include "pqd_kernels.hpp"
#include "pqd_rle_plan.hpp"
#include "pqd_whole_page.hpp"

namespace pqd {

constexpr int PLAN_THREADS = 64;
constexpr int PLAN_SLOTS = 64;       // 16-byte header blocks fetched per round, one per lane
constexpr uint32_t PLAN_NEED = 9;    // bytes a header can need: 5 (varint) + 4 (RLE value)
constexpr int32_t PLAN_ST_GENERIC = -1;  // walk status: this page is left to the generic kernel

struct PlanState {
  uint32_t pos;       // next unread stream byte
  uint32_t vals;      // values planned so far
  uint32_t pending;   // groups left of a bit-packed run whose header is consumed
  uint32_t nstage;    // segments waiting in the LDS stage
  int32_t status;
  uint32_t done;
  uint32_t need_round;  // the walk needs header bytes that the last round did not fetch
  uint32_t round_pos0;  // stream offset of block 0 of the last round
  uint32_t round_stride;
  uint32_t next_stride; // length of the last run walked: stride of the next round ...
  uint32_t prev_stride; // ... if the run before it had the same length (else the round is one contiguous KiB)
  uint32_t chunk_slot;  // pool chunk obtained for the flush in progress
  uint32_t list_n, list_after, list_stop;  // result of the chain walk of plan_window
};

// One lane.  Emits segments into `stage` until the stage is full, the walk leaves the header
// window, or the page is finished.
__device__ void plan_walk(PlanState& st, const uint32_t* win, RleSeg* stage, uint32_t w, uint32_t end,
                          uint32_t n_values, uint32_t seg_groups_cap, uint32_t seg_limit, uint32_t max_runs) {
  uint32_t pos = st.pos, vals = st.vals, pending = st.pending, ns = st.nstage;
  const uint32_t pos0 = st.round_pos0, stride = st.round_stride;
  uint32_t blk = 0, blk_pos = pos0, next_stride = st.next_stride, prev_stride = st.prev_stride;  // block that may hold the next header
  int32_t status = PQD_OK;
  bool need_round = false;
  uint32_t runs = 0;
  while (ns < seg_limit && vals < n_values) {
    const uint32_t rem = n_values - vals;
    if (pending == 0) {
      if (runs++ >= max_runs) {  // the caller wants the window parsed anew from here
        need_round = true;
        break;
      }
      if (pos >= end) {  // :81 "Reading past RLE/BitPacking stream."
        status = PQD_PAST_END;
        break;
      }
      // the header (<= 5 bytes) and an RLE value (<= 4 bytes) must lie inside a fetched block
      uint32_t off;  // LDS byte offset of stream byte `pos`
      if (stride == 16) {
        const bool inside = pos >= pos0 && (pos + PLAN_NEED <= pos0 + 16 * PLAN_SLOTS || pos0 + 16 * PLAN_SLOTS >= end);
        if (!inside) {
          need_round = true;
          break;
        }
        off = pos - pos0;
      } else {
        while (blk + 1 < (uint32_t)PLAN_SLOTS && blk_pos + stride <= pos) {
          ++blk;
          blk_pos += stride;
        }
        const bool inside = pos >= blk_pos && (pos + PLAN_NEED <= blk_pos + 16 || (blk_pos + 16 >= end && pos < blk_pos + 16));
        if (!inside) {
          need_round = true;
          break;
        }
        off = blk * 16 + (pos - blk_pos);
      }
      const uint32_t run_pos = pos;
      // BytesUtils.readUnsignedVarInt (parquet-common/.../BytesUtils.java:202-211)
      uint32_t header = 0, shift = 0, p = pos, nb = 0;
      for (;;) {
        if (p >= end) {
          status = PQD_EOF;
          break;
        }
        if (++nb > 5) {  // a header beyond five bytes may leave the fetched block: the generic kernel reads it
          status = PLAN_ST_GENERIC;
          break;
        }
        const uint32_t b = lds_byte(win, off + (p - run_pos));
        ++p;
        if (b & 0x80u) {
          header |= (b & 0x7Fu) << (shift & 31u);
          shift += 7;
        } else {
          header |= b << (shift & 31u);
          break;
        }
      }
      if (status) break;
      if ((header & 1u) == 0) {
        // RLE run; the value is ceil(w/8) little-endian bytes, not masked (BytesUtils.java:124-142)
        const uint32_t count = header >> 1;
        const uint32_t nbytes = (w + 7) >> 3;
        if (end - p < nbytes) {
          status = PQD_EOF;
          break;
        }
        uint32_t value = 0;
        for (uint32_t i = 0; i < nbytes; ++i) value |= lds_byte(win, off + (p - run_pos) + i) << (8 * i);
        pos = p + nbytes;
        prev_stride = next_stride;
        next_stride = pos - run_pos;
        // a zero-count run leaves currentCount negative: its value repeats for the rest of the page
        const uint32_t take = (count == 0 || count > rem) ? rem : count;
        stage[ns].start4 = vals << 2;
        stage[ns].src = value;
        ++ns;
        vals += take;
        continue;
      }
      const uint32_t groups = header >> 1;
      if (groups == 0 || groups > 0x0FFFFFFFu) {  // empty buffer indexed / negative array size
        status = PQD_DECODE;
        break;
      }
      pending = groups;
      pos = p;
      const uint64_t run_len = (uint64_t)groups * w + (p - run_pos);
      prev_stride = next_stride;
      next_stride = run_len > 0x40000000ull ? 0x40000000u : (uint32_t)run_len;
    }
    // ---- a piece of the current bit-packed run
    const uint32_t want = (rem + 7) >> 3;
    uint32_t g = pending < want ? pending : want;
    const uint32_t avail = end > pos ? end - pos : 0u;
    bool zeros = (w == 0);
    if (!zeros) {
      if (g > seg_groups_cap) g = seg_groups_cap;
      if ((uint64_t)g * w > avail) {
        const uint32_t with_data = (avail + w - 1) / w;  // groups holding at least one real byte
        if (with_data == 0)
          zeros = true;  // the stream ended inside the run: the rest of its buffer stays zero (:95-99)
        else
          g = with_data;
      }
    }
    if (zeros) {
      g = pending < want ? pending : want;
      const uint32_t take = g * 8 > rem ? rem : g * 8;
      stage[ns].start4 = vals << 2;
      stage[ns].src = 0;
      ++ns;
      vals += take;
      pending = 0;
      continue;
    }
    const uint32_t bytes = g * w;
    const uint32_t take = g * 8 > rem ? rem : g * 8;
    stage[ns].start4 = (vals << 2) | SEG_PACKED_BIT | (bytes > avail ? SEG_PARTIAL_BIT : 0u);
    stage[ns].src = pos;
    ++ns;
    vals += take;
    pos += bytes > avail ? avail : bytes;
    pending -= g;
  }
  if (status != PQD_OK) {
    st.status = status;
    st.done = 1;
  } else if (vals >= n_values) {
    if (pending > 0 && w > 0) {  // readNext consumed the whole payload when it read the header
      const uint64_t rest = (uint64_t)pending * w;
      const uint32_t avail = end > pos ? end - pos : 0u;
      pos += rest > avail ? avail : (uint32_t)rest;
    }
    st.done = 1;
  }
  st.pos = pos;
  st.vals = vals;
  st.pending = pending;
  st.nstage = ns;
  st.need_round = need_round ? 1u : 0u;
  st.next_stride = next_stride;
  st.prev_stride = prev_stride;
}

// A candidate run header at some byte of the header window, decoded from the eight bytes `x` that
// start there.  `len` = bytes to the next header, `nvals` = values the run yields, `src` = the
// repeated value (RLE) or the offset of the payload from the header's first byte (bit-packed).
struct Cand {
  uint32_t len, nvals, src;
  bool packed, complex;  // complex: not for the parallel path (the serial walk knows all the rules)
};

__device__ __forceinline__ Cand parse_cand(uint64_t x, uint32_t w, uint32_t vb) {
  Cand c;
  const uint32_t lo = (uint32_t)x;
  const uint32_t c0 = (lo >> 7) & 1u, c1 = (lo >> 15) & 1u, c2 = (lo >> 23) & 1u, c3 = lo >> 31;
  const uint32_t nb = 1 + c0 + (c0 & c1) + (c0 & c1 & c2);  // bytes of the varint header, if at most four
  uint32_t h = (lo & 0x7Fu) | (((lo >> 8) & 0x7Fu) << 7) | (((lo >> 16) & 0x7Fu) << 14) | (((lo >> 24) & 0x7Fu) << 21);
  h &= (1u << (7 * nb)) - 1u;
  const uint32_t cnt = h >> 1;
  c.packed = (h & 1u) != 0;
  c.complex = (c0 & c1 & c2 & c3) != 0 || cnt == 0;  // a longer header; a zero-count RLE run (repeats for ever) / empty packed run
  if (c.packed) {
    const uint64_t bytes = (uint64_t)cnt * w;
    c.complex = c.complex || bytes > (uint64_t)PLAN_SEG_MAX_BYTES || cnt > 0x0FFFFFFFu;
    c.len = nb + (uint32_t)bytes;
    c.nvals = cnt * 8;
    c.src = nb;
    if (w == 0) {  // no payload: eight zeros per group
      c.packed = false;
      c.src = 0;
    }
  } else {
    c.len = nb + vb;
    c.nvals = cnt;
    const uint64_t v = x >> (8 * nb);
    c.src = vb == 0 ? 0u : (uint32_t)(v & (vb >= 4 ? 0xFFFFFFFFull : ((1ull << (8 * vb)) - 1ull)));  // not masked to w bits
  }
  return c;
}

constexpr int PLAN_WINDOW = 16 * PLAN_SLOTS;  // bytes of a contiguous round
constexpr uint32_t NXT_COMPLEX = 0xFFFFu, NXT_STOP = 0xFFFEu;

// Parallel planning of a contiguous header window (all lanes).  Every lane decodes the sixteen
// candidate headers that start at its own bytes and writes where each would lead (`nxt`).  The chain
// of real headers from the stream position is then found in three steps: every lane resolves, for
// each of its sixteen bytes, where a chain entering there leaves the lane's block (`exitp`, back to
// front, one LDS read per byte); lane 0 hops from block to block (at most 64 hops, whatever the
// number of runs) and notes where the chain enters each block; every lane lists the runs of its
// own block from that entry, and a wave prefix sum puts the lists one behind the other.  Then the
// runs are turned into segments 64 at a time, their output positions by a wave prefix sum.  Runs
// the parallel rules do not cover (see parse_cand; also anything that touches the end of the
// stream) stop the chain: the serial walk takes that one run and the next round starts behind it.
// `exitp` (1024 entries) doubles as the lanes' private lists; `entry_of` has one entry per lane.
__device__ __forceinline__ void plan_window(PlanState& st, const u32x4* win4, uint16_t* nxt, uint16_t* list, uint16_t* exitp,
                                            uint16_t* entry_of, RleSeg* stage, uint32_t end, uint32_t w, uint32_t n_values,
                                            uint32_t seg_limit, uint32_t tid) {
  const uint32_t pos0 = st.round_pos0;
  const uint32_t entry = st.pos - pos0;
  const uint32_t vb = (w + 7) >> 3;
  const bool to_end = (uint64_t)pos0 + PLAN_WINDOW >= end;  // the window holds the rest of the stream
  const uint32_t block_end = 16 * tid + 16;
  // ---- A. candidates
  {
    const u32x4 a = win4[tid], b = win4[tid + 1];  // win4[PLAN_SLOTS] is zero
    const uint32_t d[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
    uint16_t out[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t p = 16 * tid + j;
      const uint32_t sh = (j & 3) * 8;
      const uint32_t x0 = funnel_shr(d[j >> 2], d[(j >> 2) + 1], sh), x1 = funnel_shr(d[(j >> 2) + 1], d[(j >> 2) + 2 < 6 ? (j >> 2) + 2 : 5], sh);
      const Cand c = parse_cand(((uint64_t)x1 << 32) | x0, w, vb);
      uint32_t v = p + c.len;  // < 1024 + 4 + PLAN_SEG_MAX_BYTES
      if (c.complex || (uint64_t)pos0 + p + c.len > end) v = NXT_COMPLEX;
      if (p + PLAN_NEED > (uint32_t)PLAN_WINDOW && !to_end) v = NXT_STOP;  // its bytes may lie outside the window
      if ((uint64_t)pos0 + p >= end) v = NXT_COMPLEX;                      // a header at the end of the stream: the serial walk reports it
      out[j] = (uint16_t)v;
    }
    u32x4* dst = reinterpret_cast<u32x4*>(nxt + 16 * tid);
    dst[0] = u32x4{(uint32_t)out[0] | ((uint32_t)out[1] << 16), (uint32_t)out[2] | ((uint32_t)out[3] << 16),
                   (uint32_t)out[4] | ((uint32_t)out[5] << 16), (uint32_t)out[6] | ((uint32_t)out[7] << 16)};
    dst[1] = u32x4{(uint32_t)out[8] | ((uint32_t)out[9] << 16), (uint32_t)out[10] | ((uint32_t)out[11] << 16),
                   (uint32_t)out[12] | ((uint32_t)out[13] << 16), (uint32_t)out[14] | ((uint32_t)out[15] << 16)};
    // ---- B1. where a chain that enters at byte j leaves this block: a position at or past the
    // block's end, or the position inside the block at which the chain stops.  A candidate leads
    // forward, so the entries behind j are known when j's turn comes.
#pragma unroll
    for (int j = 15; j >= 0; --j) {
      const uint32_t q = out[j];
      uint32_t e = q;
      if (q >= NXT_STOP)
        e = 16 * tid + j;
      else if (q < block_end)
        e = exitp[q];
      exitp[16 * tid + j] = (uint16_t)e;
    }
    entry_of[tid] = 0xFFFFu;
  }
  __syncthreads();
  // ---- B2. the chain from the stream position, block by block
  if (tid == 0) {
    uint32_t p = entry, stop = NXT_STOP;
    while (p < (uint32_t)PLAN_WINDOW) {
      const uint32_t blk = p >> 4;
      entry_of[blk] = (uint16_t)p;
      const uint32_t e = exitp[p];
      if (e < 16 * blk + 16) {  // the chain stops inside this block
        p = e;
        stop = nxt[e];
        break;
      }
      p = e;
    }
    st.list_after = p;     // window offset behind the last run listed
    st.list_stop = stop;   // NXT_COMPLEX: the run at list_after is for the serial walk
  }
  __syncthreads();
  // ---- B3. the runs of every block the chain visits, then all of them in stream order
  {
    uint16_t* mine = exitp + 16 * tid;  // B2 is done with this block's exits
    uint32_t c = 0, p = entry_of[tid];
    while (p < block_end) {  // 0xFFFF: the chain does not come through here
      const uint32_t q = nxt[p];
      if (q >= NXT_STOP) break;
      mine[c++] = (uint16_t)p;
      p = q;
    }
    uint32_t inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = __shfl_up(inc, off, 64);
      if (tid >= (uint32_t)off) inc += up;
    }
    for (uint32_t i = 0; i < c; ++i) list[inc - c + i] = mine[i];
    if (tid == PLAN_THREADS - 1) st.list_n = inc;
  }
  __syncthreads();
  // ---- C. segments, 64 runs at a time
  const uint32_t n = st.list_n;
  const uint32_t* wb = reinterpret_cast<const uint32_t*>(win4);
  uint32_t vals = st.vals, ns = st.nstage, taken = 0;
  bool finished = false, full = false;
  for (uint32_t base = 0; base < n && !finished && !full; base += PLAN_THREADS) {
    const bool have = base + tid < n;
    const uint32_t p = have ? list[base + tid] : 0u;
    const uint32_t i0 = p >> 2, sh = (p & 3u) * 8;
    const uint32_t x0 = funnel_shr(wb[i0], wb[i0 + 1], sh), x1 = funnel_shr(wb[i0 + 1], wb[i0 + 2], sh);
    const Cand c = parse_cand(((uint64_t)x1 << 32) | x0, w, vb);
    const uint32_t cap = n_values - vals;  // > 0 here
    const uint32_t mine = have ? (c.nvals < cap ? c.nvals : cap) : 0u;
    // inclusive prefix sum over the wave, saturating at cap (all operands <= 2^30)
    uint32_t inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = __shfl_up(inc, off, 64);
      if (tid >= (uint32_t)off) {
        inc += up;
        inc = inc < cap ? inc : cap;
      }
    }
    const uint32_t prev_inc = __shfl_up(inc, 1, 64);
    const uint32_t before = tid == 0 ? 0u : prev_inc;  // values of the runs before this one, saturated at cap
    const bool live = have && before < cap && ns + tid < seg_limit;  // starts inside the page and fits the stage
    const uint64_t lm = __builtin_amdgcn_ballot_w64(live);
    const uint32_t m = lm == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~lm);  // leading live runs
    if (tid < m) {
      stage[ns + tid].start4 = ((vals + before) << 2) | (c.packed ? SEG_PACKED_BIT : 0u);
      stage[ns + tid].src = c.packed ? pos0 + p + c.src : c.src;
    }
    const uint32_t have_n = n - base < (uint32_t)PLAN_THREADS ? n - base : (uint32_t)PLAN_THREADS;
    const uint32_t total = m ? __shfl(inc, (int)m - 1, 64) : 0u;  // values of the runs taken (saturated)
    vals += total;
    ns += m;
    taken += m;
    if (vals >= n_values) finished = true;
    if (m < have_n) full = !finished;  // stopped for lack of stage room
  }
  if (tid == 0) {
    st.vals = vals;
    st.nstage = ns;
    const uint32_t after = taken < n ? (uint32_t)list[taken] : st.list_after;
    if (taken) {
      st.pos = pos0 + after;
      // the lengths of the last two runs taken: what decides between a strided and a contiguous round
      const uint32_t last = after - list[taken - 1];
      st.prev_stride = taken >= 2 ? (uint32_t)list[taken - 1] - list[taken - 2] : st.next_stride;
      st.next_stride = last;
    }
    if (finished) st.done = 1;
    // a complex run right behind the ones taken goes to the serial walk with this window; otherwise fetch anew
    st.need_round = (!finished && !full && taken == n && st.list_stop == NXT_COMPLEX) ? 0u : 1u;
  }
  __syncthreads();
}

// One fetch round (all lanes): lane l loads the 16 bytes at pos + l*stride into the header window
// (bytes past the stream end read as zero).  When the stride is a run length (the last run walked
// was a bit-packed run of `stride` bytes), the lanes then try the regular case in parallel: every
// lane parses the header in its own block, and the leading lanes whose run is a full bit-packed
// run of that same shape (same header bytes, same length, wholly inside the stream, all of its
// values wanted) emit their segments at once.  Pages as parquet-java writes them (63-group runs
// behind 1-byte headers) are planned 64 runs per round this way; anything else is left to the
// serial walk, which starts where the regular prefix ends.
__device__ __forceinline__ void plan_round(PlanState& st, u32x4* win4, RleSeg* stage, const uint8_t* data, uint32_t end,
                                           uint32_t w, uint32_t n_values, uint32_t seg_limit, uint32_t tid) {
  const uint32_t cur = st.pos;
  const uint32_t stride = (st.next_stride >= 24 && st.next_stride == st.prev_stride) ? st.next_stride : 16u;
  const uint64_t at = (uint64_t)cur + (uint64_t)tid * stride;
  u32x4 v = {0u, 0u, 0u, 0u};
  if (at + 16 <= end) {
    __builtin_memcpy(&v, data + at, 16);
  } else if (at < end) {
    uint32_t tmp[4] = {0u, 0u, 0u, 0u};
    for (uint32_t i = 0; i < (uint32_t)(end - at); ++i) tmp[i >> 2] |= (uint32_t)data[at + i] << (8 * (i & 3u));
    v = u32x4{tmp[0], tmp[1], tmp[2], tmp[3]};
  }
  const uint32_t vals = st.vals, ns = st.nstage, pending = st.pending;
  __syncthreads();
  win4[tid] = v;
  if (tid == 0) {
    win4[PLAN_SLOTS] = u32x4{0u, 0u, 0u, 0u};
    st.round_pos0 = cur;
    st.round_stride = stride;
  }
  // ---- the regular case
  uint32_t m = 0;
  if (stride >= 24 && pending == 0 && w > 0) {
    // varint header in the first bytes of the block (at most 4 bytes here: longer ones go the serial way)
    const uint32_t b0 = v.x & 0xFFu, b1 = (v.x >> 8) & 0xFFu, b2 = (v.x >> 16) & 0xFFu, b3 = v.x >> 24;
    uint32_t header, nb;
    if (!(b0 & 0x80u)) {
      header = b0;
      nb = 1;
    } else if (!(b1 & 0x80u)) {
      header = (b0 & 0x7Fu) | (b1 << 7);
      nb = 2;
    } else if (!(b2 & 0x80u)) {
      header = (b0 & 0x7Fu) | ((b1 & 0x7Fu) << 7) | (b2 << 14);
      nb = 3;
    } else {
      header = (b0 & 0x7Fu) | ((b1 & 0x7Fu) << 7) | ((b2 & 0x7Fu) << 14) | (b3 << 21);
      nb = (b3 & 0x80u) ? 0u : 4u;
    }
    const uint32_t groups = header >> 1;
    const uint32_t h0 = __builtin_amdgcn_readfirstlane(header), nb0 = __builtin_amdgcn_readfirstlane(nb);
    const uint32_t g0 = h0 >> 1;
    // lane 0 decides the shape; a lane fits if its run has that shape and lies inside the stream and the page
    const bool shape = nb0 != 0 && (h0 & 1u) && g0 >= 1 && (uint64_t)g0 * w + nb0 == stride &&
                       (uint64_t)g0 * w <= (uint64_t)PLAN_SEG_MAX_BYTES;
    const bool fits = shape && header == h0 && nb == nb0 && groups == g0 && at + stride <= end &&
                      (uint64_t)vals + (uint64_t)(tid + 1) * g0 * 8 <= n_values && ns + tid < seg_limit;
    const uint64_t miss = ~__builtin_amdgcn_ballot_w64(fits);
    m = miss ? (uint32_t)__builtin_ctzll(miss) : 64u;
    if (tid < m) {
      stage[ns + tid].start4 = ((vals + tid * g0 * 8) << 2) | SEG_PACKED_BIT;
      stage[ns + tid].src = (uint32_t)at + nb0;
    }
    if (tid == 0 && m) {
      st.pos = cur + m * stride;
      st.vals = vals + m * g0 * 8;
      st.nstage = ns + m;
      if (st.vals >= n_values) st.done = 1;
    }
  }
  if (tid == 0) st.need_round = (m == 64u) ? 1u : 0u;  // a whole round of regular runs: the next one starts past it
  __syncthreads();
}

// Pool chunk p is reserved for page p; a page that has nothing for the consuming kernel leaves an
// empty chunk there (v0 == v1), so that the consumers can take every slot as it comes.
__device__ __forceinline__ void write_empty_chunk(RleSeg* pool, uint32_t slot, uint32_t page) {
  RleChunkHead hd = {};
  hd.data = (uint64_t)(uintptr_t)pool;  // any readable address
  hd.page = page;
  *reinterpret_cast<RleChunkHead*>(pool + (size_t)slot * SEG_CHUNK + SEG_HEAD_SLOT) = hd;
}
__device__ __forceinline__ void write_empty_chunk(RleSeg* pool, uint32_t page) { write_empty_chunk(pool, page, page); }

constexpr int BULK_LOADS = 8;  // run headers a lane checks per step of the bulk verification

// Varint of at most four bytes from the dword that starts with it (nb = 0: longer than that).
__device__ __forceinline__ void varint_of_dword(uint32_t x, uint32_t& header, uint32_t& nb) {
  const uint32_t b0 = x & 0xFFu, b1 = (x >> 8) & 0xFFu, b2 = (x >> 16) & 0xFFu, b3 = x >> 24;
  if (!(b0 & 0x80u)) {
    header = b0;
    nb = 1;
  } else if (!(b1 & 0x80u)) {
    header = (b0 & 0x7Fu) | (b1 << 7);
    nb = 2;
  } else if (!(b2 & 0x80u)) {
    header = (b0 & 0x7Fu) | ((b1 & 0x7Fu) << 7) | (b2 << 14);
    nb = 3;
  } else {
    header = (b0 & 0x7Fu) | ((b1 & 0x7Fu) << 7) | ((b2 & 0x7Fu) << 14) | (b3 << 21);
    nb = (b3 & 0x80u) ? 0u : 4u;
  }
}

// Key and first probe position of a dictionary in the plane-slot table (pqd_rle_plan.hpp).  Ids of
// planned pages are below 65536 whatever the dictionary's length: entries beyond that are never read.
__device__ __forceinline__ uint64_t dict_plane_key(const void* dict, uint32_t dict_len) {
  const uint32_t entries = dict_len < 65536u ? dict_len : 65536u;
  return ((uint64_t)(uintptr_t)dict & 0xFFFFFFFFFFFFull) | ((uint64_t)(entries - 1u) << 48);
}
__device__ __forceinline__ uint32_t dict_plane_hash(uint64_t key, uint32_t n_slots) {
  return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) % n_slots;
}

// Claims a plane slot for every 8-byte dictionary of the batch: one lane per page; a page whose
// dictionary is that of the page before it has nothing to do, so a column chunk's pages cost one
// table access (the plan kernel then only reads the table: thousands of pages asking for the same
// entry with atomics took longer than planning them).  Whoever puts a key into the table also lists
// the slot for the kernel that fills it.
__global__ __launch_bounds__(256) void k_dict_slots(const pqd_page_desc* __restrict__ descs, uint32_t n_pages,
                                                    uint32_t* __restrict__ cursor, DictPlanes planes) {
  const uint32_t page = blockIdx.x * 256u + threadIdx.x;
  if (page >= n_pages || planes.n_slots == 0) return;
  const pqd_page_desc d = descs[page];
  if (d.encoding != PQD_ENC_RLE_DICTIONARY || d.elem_size != 8 || d.dict_len == 0 || ((uintptr_t)d.dict & 15u) != 0) return;
  if (page > 0) {
    const pqd_page_desc b = descs[page - 1];
    if (b.dict == d.dict && b.dict_len == d.dict_len && b.encoding == d.encoding && b.elem_size == d.elem_size) return;
  }
  const uint64_t key = dict_plane_key(d.dict, d.dict_len);
  uint32_t h = dict_plane_hash(key, planes.n_slots);
  const uint32_t probes = planes.n_slots < DP_MAX_PROBES ? planes.n_slots : DP_MAX_PROBES;
  for (uint32_t i = 0; i < probes; ++i) {
    unsigned long long* e = reinterpret_cast<unsigned long long*>(planes.keys + h);
    unsigned long long cur = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0ull) {
      cur = atomicCAS(e, 0ull, (unsigned long long)key);
      if (cur == 0ull) {
        planes.list[atomicAdd(cursor + 2, 1u)] = h;
        return;
      }
    }
    if (cur == key) return;
    h = h + 1u == planes.n_slots ? 0u : h + 1u;
  }
}

// ---------------------------------------------------------------------------------------------
// The plan kernel: one wave per page, pages of any length.  The plan is a list of chunks taken from
// a shared pool (pqd_rle_plan.hpp): plain hybrid pages (want_dict == 0: levels, booleans, ids as
// ints) are consumed by k_rle_expand, dictionary pages (want_dict == 1) by k_dict_pages_sliced,
// whose workgroup holds a whole chunk in registers (hence the smaller span).  The page's result
// is written here; the dictionary kernel amends it if it meets an id beyond the dictionary.
__global__ __launch_bounds__(PLAN_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void k_rle_plan_chunks(const pqd_page_desc* __restrict__ descs,
                                                                  pqd_page_result* __restrict__ results,
                                                                  RlePagePlan* __restrict__ plans, RleSeg* __restrict__ pool,
                                                                  uint32_t pool_chunks, uint32_t* __restrict__ cursor,
                                                                  uint32_t n_pages, int want_dict, int elem,
                                                                  DictPlanes planes, int small_ran, uint32_t direct_max,
                                                                  uint32_t whole_max, int long_ok) {
  __shared__ u32x4 win4[PLAN_SLOTS + 1];
  __shared__ RleSeg stage[SEG_CHUNK];
  __shared__ RleSeg chunk[SEG_CHUNK];
  __shared__ u32x4 nxt4[PLAN_WINDOW / 8], list4[PLAN_WINDOW / 8];
  __shared__ PlanState st;
  __shared__ uint16_t entry_of[PLAN_THREADS];
  uint16_t* nxt = reinterpret_cast<uint16_t*>(nxt4);
  uint16_t* list = reinterpret_cast<uint16_t*>(list4);
  const uint32_t* win = reinterpret_cast<const uint32_t*>(win4);
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const uint32_t tid = threadIdx.x;
  // the plane copies of the 8-byte dictionaries k_dict_slots claimed slots for (the gather kernel reads them; nothing
  // in this kernel does): a piece per workgroup, ahead of its page, instead of a launch of its own
  if (want_dict && elem == 8 && planes.n_slots != 0) {
    uint32_t n = __builtin_amdgcn_readfirstlane(cursor[PLAN_CTR_PLANES]);
    if (n > planes.n_slots) n = planes.n_slots;
    for (uint32_t item = page; item < n * DP_PIECES; item += n_pages) dict_plane_piece(planes, item, tid, PLAN_THREADS);
  }
  const pqd_page_desc d = descs[page];
  if (small_ran && plans[page].flags == PLAN_F_SMALL_DONE) {  // k_rle_small_pages has decoded this page
    if (tid == 0) {
      plans[page].flags = PLAN_F_DONE;
      if (page < pool_chunks) write_empty_chunk(pool, page);
    }
    return;
  }

  // The page's first eight bytes (prefix and first run header) and the first probe of the plane-slot table are
  // requested together, before anything waits: a page is planned by one wave, and every dependent trip to
  // memory is a microsecond of the kernel.
  uint32_t first8[2] = {0u, 0u};
  const bool have8 = d.data_len >= 8u && d.data != nullptr;
  if (have8) __builtin_memcpy(first8, d.data, 8);
  const bool probe_early = want_dict && elem == 8 && planes.n_slots != 0 && d.dict_len != 0;
  const uint64_t key_early = probe_early ? dict_plane_key(d.dict, d.dict_len) : 0ull;
  const uint32_t h_early = probe_early ? dict_plane_hash(key_early, planes.n_slots) : 0u;
  const uint64_t cur_early = probe_early ? planes.keys[h_early] : 0ull;
  const uint64_t first64 = ((uint64_t)first8[1] << 32) | first8[0];

  // ---- validation and stream prefixes (same rules as the generic kernel)
  int32_t early = PQD_OK;
  bool finished = false;
  uint32_t pos = 0, end = d.data_len, w = d.bit_width, consumed_override = ~0u;
  const uint32_t n_values = d.num_values;
  const bool is_dict = d.encoding == PQD_ENC_RLE_DICTIONARY;
  if ((d.encoding != PQD_ENC_RLE && !is_dict) || (int)is_dict != want_dict || (int)d.elem_size != elem) {
    early = PQD_BAD_ARG;
    finished = true;
  } else if (is_dict || (d.flags & PQD_F_WIDTH_PREFIX)) {
    if (end == 0) {
      early = n_values ? PQD_DECODE : PQD_OK;
      finished = true;
    } else {
      w = have8 ? (uint32_t)(first64 & 0xFFu) : d.data[0];
      pos = 1;
    }
  } else if (d.flags & PQD_F_LENGTH_PREFIX) {
    if (end < 4) {
      early = PQD_EOF;
      finished = true;
    } else {
      const uint32_t len = have8 ? (uint32_t)first64
                                 : (uint32_t)d.data[0] | ((uint32_t)d.data[1] << 8) | ((uint32_t)d.data[2] << 16) | ((uint32_t)d.data[3] << 24);
      if ((int32_t)len < 0) {
        early = PQD_DECODE;
        finished = true;
      } else if (len > end - 4) {
        early = PQD_EOF;
        finished = true;
      } else {
        pos = 4;
        end = 4 + len;
        consumed_override = end;
      }
    }
  }
  if (!finished && w > 32) {
    early = PQD_BAD_ARG;
    finished = true;
  }
  if (finished || n_values == 0) {
    if (tid == 0) {
      pqd_page_result r;
      r.status = early;
      r.values_written = 0;
      r.bytes_consumed = early == PQD_OK ? (consumed_override != ~0u ? consumed_override : pos) : 0;
      r.reserved = 0;
      results[page] = r;
      plans[page].flags = PLAN_F_DONE;
      if (page < pool_chunks) write_empty_chunk(pool, page);
    }
    return;
  }
  // Run-heavy dictionary pages (a short stream for their number of values) are not planned at all: the left-over launch
  // decodes them whole, ids and lookups in one pass (pqd_whole_page.hpp), and falls back to its page decoder by itself.
  // Long pages of many short runs, when the launch sequence has the kernel for them (few pages in the call: a
  // workgroup of sixteen waves per page then has a CU to itself): one wave planning 25000 runs is what such a call waits for.
  if (long_ok && (!want_dict || d.dict != nullptr) && whole_page_long_candidate(end - pos, n_values, w, want_dict ? d.dict_len : 1u, whole_max)) {
    if (tid == 0) {
      plans[page].flags = PLAN_F_LONG;
      if (page < pool_chunks) write_empty_chunk(pool, page);
    }
    return;
  }
  if (want_dict && d.dict != nullptr && whole_page_candidate(end - pos, n_values, w, d.dict_len, whole_max)) {
    if (tid == 0) {
      plans[page].flags = PLAN_F_WHOLE;
      cursor[PLAN_CTR_WHOLE] = 1u;
      if (page < pool_chunks) write_empty_chunk(pool, page);
    }
    return;
  }
  // The same for run-heavy pages of dictionary ids that are read as ints (a width byte in front, no dictionary).
  if (!want_dict && (d.flags & PQD_F_WIDTH_PREFIX) && whole_page_candidate(end - pos, n_values, w, 1u, whole_max)) {
    if (tid == 0) {
      plans[page].flags = PLAN_F_WHOLE;
      cursor[PLAN_CTR_FALLBACK] = 1u;  // (the left-over launch of plain pages looks at this flag only)
      if (page < pool_chunks) write_empty_chunk(pool, page);
    }
    return;
  }
  // Not for the planned kernels: pages of more than 2^30 values (segment starts hold 30 bits);
  // dictionary pages with ids wider than 16 bits, or whose dictionary is misaligned or too large
  // for streaming it through LDS to be cheaper than a direct gather.
  bool not_planned = n_values > SEG_START_MAX;
  if (want_dict) {
    const uint32_t per_slice = DS_SLICE_BYTES / (uint32_t)elem;
    const uint32_t passes = (d.dict_len + per_slice - 1) / per_slice;
    const uint32_t held = n_values < DS_TILE_VALUES ? n_values : DS_TILE_VALUES;
    const bool sweepable = ((uintptr_t)d.dict & 15u) == 0 &&
                           (passes <= 1 || (passes <= DS_MAX_PASSES && (uint64_t)d.dict_len * (uint32_t)elem <= 32ull * held));
    not_planned = not_planned || w > 16 || !sweepable;
  }
  // 8-byte dictionaries are looked up in plane copies (pqd_rle_plan.hpp): the slot of this page's
  // dictionary, claimed by k_dict_slots before this kernel ran.  No slot (table or area full): the
  // page is left to the generic kernel, which reads the dictionary from memory.
  uint32_t slot1 = 0;  // 1 + slot
  const bool needs_planes = want_dict && elem == 8 && d.dict_len != 0 && !not_planned;
  if (needs_planes && planes.n_slots != 0) {
    const uint64_t key = dict_plane_key(d.dict, d.dict_len);
    uint32_t h = dict_plane_hash(key, planes.n_slots);
    const uint32_t probes = planes.n_slots < DP_MAX_PROBES ? planes.n_slots : DP_MAX_PROBES;
    for (uint32_t i = 0; i < probes && slot1 == 0; ++i) {
      const uint64_t cur = i == 0 && probe_early ? cur_early : planes.keys[h];
      if (cur == key) slot1 = h + 1u;
      if (cur == 0ull) break;  // a free entry ends the probe sequence: the dictionary has no slot
      h = h + 1u == planes.n_slots ? 0u : h + 1u;
    }
    slot1 = __builtin_amdgcn_readfirstlane(slot1);
  }
  not_planned = not_planned || (needs_planes && slot1 == 0);
  if (not_planned) {
    if (tid == 0) {
      plans[page].flags = PLAN_F_FALLBACK;
      cursor[PLAN_CTR_FALLBACK] = 1u;  // (a flag: the left-over launch then looks through the plan records)
      if (page < pool_chunks) write_empty_chunk(pool, page);
    }
    return;
  }
  const uint32_t nseg_tag = slot1 << 16;
  if (tid == 0) {
    st.pos = pos;
    st.vals = 0;
    st.pending = 0;
    st.nstage = 0;
    st.status = PQD_OK;
    st.done = 0;
    st.need_round = 1;
    st.round_pos0 = pos;
    st.round_stride = 16;
    st.next_stride = 16;
    st.prev_stride = 16;
    st.chunk_slot = 0;
  }
  __syncthreads();

  // dictionaries the regular path of k_dict_pages_sliced takes: resident in LDS, or swept in four quarters per plane
  // (4-byte dictionaries only for the resident kind: see gather_regular_batch)
  const bool regular_dictionary = want_dict && (d.dict_len > 3u * DP_QUARTER_ENTRIES || (elem == 4 && d.dict_len <= 2u * DP_QUARTER_ENTRIES));
  const uint32_t front_limit = plan_front_limit(pool_chunks, n_pages, want_dict != 0);  // slots below it: page slots + front of the tail
  const uint32_t seg_groups_cap = w ? (uint32_t)PLAN_SEG_MAX_BYTES / w : 1u;
  uint32_t v0 = 0;  // start of the chunk being collected
  uint32_t emitted = 0;  // chunks written for this page
  const uint32_t span = want_dict ? DS_TILE_VALUES : SEG_CHUNK_SPAN;  // values per chunk, at most
  bool fallback = false;

  // ---- pages that start with equal bit-packed runs (non-repeating data as parquet-java writes it:
  // 63-group runs behind 1-byte headers).  Where run r starts follows from r alone, so the lanes
  // check the headers of all runs, BULK_LOADS each per step.  Of a long page, the chunks that lie
  // wholly inside the verified prefix are written straight from arithmetic: no staging, no searches.
  // What remains of the page (the last runs, anything irregular) goes through the loop below.
  if (w > 0 && (uint64_t)pos + 16 <= end) {
    uint32_t hx;
    if (have8 && pos <= 4)
      hx = (uint32_t)(first64 >> (8 * pos));
    else
      __builtin_memcpy(&hx, d.data + pos, 4);
    uint32_t header, nb;
    varint_of_dword(hx, header, nb);
    const uint32_t g0 = header >> 1;
    const uint64_t run_bytes = (uint64_t)g0 * w + nb;
    if (nb != 0 && (header & 1u) && g0 >= 1 && run_bytes >= 24 && (uint64_t)g0 * w <= (uint64_t)PLAN_SEG_MAX_BYTES) {
      const uint32_t S = (uint32_t)run_bytes, V = g0 * 8;
      const uint32_t hmask = nb == 4 ? ~0u : (1u << (8 * nb)) - 1u, hbytes = hx & hmask;
      uint32_t rmax = (end - pos) / S;  // runs wholly inside the stream ...
      if (n_values / V < rmax) rmax = n_values / V;  // ... all of whose values are wanted
      uint32_t reg = rmax;  // leading runs with exactly that header
      // the header of the run behind the prefix, should all rmax runs turn out equal (requested with the checks below)
      const uint64_t p2_guess = (uint64_t)pos + (uint64_t)rmax * S;
      uint32_t hx2_guess = 0;
      if (p2_guess + 4 <= end) __builtin_memcpy(&hx2_guess, d.data + p2_guess, 4);
      for (uint32_t base = 0; base < rmax; base += 64u * BULK_LOADS) {
        uint32_t x[BULK_LOADS];
#pragma unroll
        for (int k = 0; k < BULK_LOADS; ++k) {
          const uint32_t r = base + (uint32_t)k * 64u + tid;
          x[k] = hbytes;
          if (r < rmax) __builtin_memcpy(&x[k], d.data + pos + (uint64_t)r * S, 4);
        }
        uint32_t first_miss = ~0u;
#pragma unroll
        for (int k = BULK_LOADS - 1; k >= 0; --k) {
          const uint64_t miss = __builtin_amdgcn_ballot_w64((x[k] & hmask) != hbytes);
          if (miss) first_miss = base + (uint32_t)k * 64u + (uint32_t)__builtin_ctzll(miss);
        }
        if (first_miss != ~0u) {
          reg = first_miss;
          break;
        }
      }
      uint32_t sp = ((229u * V) / SEG_BLOCK) * SEG_BLOCK;  // values per chunk: at most SEG_GEN_MAX runs touch it
      if (sp > span) sp = span;
      const uint32_t vreg = reg * V;
      const uint32_t n_bulk = sp ? vreg / sp : 0u;
      if (n_bulk >= 2) {
        if (tid == 0) st.chunk_slot = atomicAdd(cursor + PLAN_CTR_TAIL, n_bulk - 1u);
        __syncthreads();
        const uint32_t tail0 = n_pages + st.chunk_slot;  // chunks 1.. of this page; chunk 0 sits at the page's own slot
        if (page >= front_limit || (uint64_t)tail0 + (n_bulk - 1u) > front_limit) {
          // the pool is too small for this page: the slots taken must still read as chunks
          for (uint32_t c = 1 + tid; c < n_bulk; c += PLAN_THREADS)
            if (tail0 + c - 1 < front_limit) write_empty_chunk(pool, tail0 + c - 1, page);
          fallback = true;
        } else {
          // The expansion kernel reads only the first segment and the header of a regular chunk: for
          // plain hybrid pages every lane writes whole chunks.  The dictionary kernel has no such path
          // and gets the full segment list and block table.
          for (uint32_t c = tid; !want_dict && c < n_bulk; c += PLAN_THREADS) {
            const uint32_t a = c * sp, b = a + sp;
            const uint32_t r_first = a / V, r_last = (b - 1u) / V;
            RleSeg* out = pool + (size_t)(c == 0 ? page : tail0 + c - 1) * SEG_CHUNK;
            RleSeg sg;
            sg.start4 = ((r_first * V) << 2) | SEG_PACKED_BIT;
            sg.src = pos + r_first * S + nb;
            out[0] = sg;
            RleChunkHead hd = {};
            hd.data = (uint64_t)(uintptr_t)d.data;
            hd.out = (uint64_t)(uintptr_t)d.out;
            hd.v0 = a;
            hd.v1 = b;
            hd.nseg = r_last - r_first + 1u;
            hd.w = w;
            hd.data_len = d.data_len;
            hd.stream_end = end;
            hd.page = page;
            hd.reserved[0] = V;
            hd.reserved[1] = S;
            *reinterpret_cast<RleChunkHead*>(out + SEG_HEAD_SLOT) = hd;
          }
          for (uint32_t c = 0; want_dict && c < n_bulk; ++c) {
            const uint32_t a = c * sp, b = a + sp;
            const uint32_t r_first = a / V, r_last = (b - 1u) / V, k = r_last - r_first + 1u;
            RleSeg* out = pool + (size_t)(c == 0 ? page : tail0 + c - 1) * SEG_CHUNK;
            for (uint32_t i = tid; i <= k; i += PLAN_THREADS) {
              const uint32_t r = r_first + i;
              RleSeg sg;
              sg.start4 = i < k ? ((r * V) << 2) | SEG_PACKED_BIT : b << 2;  // the last one is the sentinel
              sg.src = i < k ? pos + r * S + nb : 0u;
              out[i] = sg;
            }
            if (tid == 0) {
              RleChunkHead hd = {};
              hd.data = (uint64_t)(uintptr_t)d.data;
              hd.out = (uint64_t)(uintptr_t)d.out;
              hd.v0 = a;
              hd.v1 = b;
              hd.nseg = k | nseg_tag;
              hd.w = w;
              hd.data_len = d.data_len;
              hd.stream_end = end;
              hd.page = page;
              hd.dict = (uint64_t)(uintptr_t)d.dict;
              hd.dict_len = d.dict_len;
              hd.reserved[0] = V;  // marks the chunk as regular: run r0 + i = values start0 + i*V, bytes src0 + i*S
              hd.reserved[1] = S;
              *reinterpret_cast<RleChunkHead*>(out + SEG_HEAD_SLOT) = hd;
              // (the regular path of k_dict_pages_sliced takes batches of such chunks over four-quarter dictionaries only)
              if (c == 0 && regular_dictionary) cursor[d.dict_len > 3u * DP_QUARTER_ENTRIES ? PLAN_CTR_SWEPT : PLAN_CTR_RESIDENT] = 1u;
              if (c == 0 && !regular_dictionary) cursor[PLAN_CTR_IRREGULAR] = 1u  /* a flag, not a count: thousands of atomics on one word queue up behind each other */;
            }
            if (tid < (uint32_t)SEG_TABLE_BLOCKS / 4) {  // segment of every 256th value, four entries per lane
              uint32_t packed = 0;
#pragma unroll
              for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t first = a + (tid * 4 + q) * SEG_BLOCK;
                const uint32_t idx = first >= b ? k - 1u : first / V - r_first;
                packed |= idx << (8 * q);
              }
              reinterpret_cast<uint32_t*>(out + SEG_TABLE_SLOT)[tid] = packed;
            }
          }
          // the runs of the prefix that reach past the last bulk chunk stay staged
          const uint32_t cut = n_bulk * sp;
          const uint32_t r_rem = cut < vreg ? cut / V : reg;
          for (uint32_t i = tid; i < reg - r_rem; i += PLAN_THREADS) {
            stage[i].start4 = (((r_rem + i) * V) << 2) | SEG_PACKED_BIT;
            stage[i].src = pos + (r_rem + i) * S + nb;
          }
          if (tid == 0) {
            st.pos = pos + reg * S;
            st.vals = vreg;
            st.nstage = reg - r_rem;
            st.done = vreg >= n_values ? 1u : 0u;
          }
          v0 = cut;
          emitted = n_bulk;
        }
        __syncthreads();
      } else if (reg >= 2) {
        // a short page: its leading equal runs go to the stage as they are, the loop below cuts them
        const uint32_t take = reg < (uint32_t)SEG_GEN_MAX ? reg : (uint32_t)SEG_GEN_MAX;
        for (uint32_t i = tid; i < take; i += PLAN_THREADS) {
          stage[i].start4 = ((i * V) << 2) | SEG_PACKED_BIT;
          stage[i].src = pos + i * S + nb;
        }
        if (tid == 0) {
          st.pos = pos + take * S;
          st.vals = take * V;
          st.nstage = take;
          st.done = take * V >= n_values ? 1u : 0u;
          if (take < reg) st.next_stride = st.prev_stride = S;  // more of the same follows: strided rounds
        }
        __syncthreads();
      }
      // The run behind the prefix: pages of non-repeating data end with one shorter bit-packed run.
      // If it lies wholly inside the stream and yields all the values still wanted, it is staged
      // here and the page is done (the loop below then only cuts the chunks).
      const bool prefix_staged = n_bulk >= 2 ? !fallback : (reg >= 2 && reg <= (uint32_t)SEG_GEN_MAX);
      const uint64_t p2 = (uint64_t)pos + (uint64_t)reg * S;
      if (prefix_staged && vreg < n_values && p2 + 4 <= end) {
        uint32_t hx2, header2, nb2;
        if (p2 == p2_guess)
          hx2 = hx2_guess;
        else
          __builtin_memcpy(&hx2, d.data + p2, 4);
        varint_of_dword(hx2, header2, nb2);
        const uint32_t g2 = header2 >> 1;
        const uint32_t ns = st.nstage;
        if (nb2 != 0 && (header2 & 1u) && g2 >= 1 && (uint64_t)g2 * w <= (uint64_t)PLAN_SEG_MAX_BYTES &&
            p2 + nb2 + (uint64_t)g2 * w <= end && (uint64_t)vreg + (uint64_t)g2 * 8 >= n_values && ns < (uint32_t)SEG_GEN_MAX) {
          __syncthreads();
          if (tid == 0) {
            stage[ns].start4 = (vreg << 2) | SEG_PACKED_BIT;
            stage[ns].src = (uint32_t)p2 + nb2;
            st.nstage = ns + 1;
            st.vals = n_values;
            st.pos = (uint32_t)p2 + nb2 + g2 * w;  // readNext takes the whole run when it reads the header
            st.done = 1;
          }
          __syncthreads();
        }
      }
    }
  }

  for (; !fallback;) {
    bool fresh = false;
    if (st.need_round && !st.done) {
      plan_round(st, win4, stage, d.data, end, w, n_values, (uint32_t)SEG_GEN_MAX, tid);
      fresh = true;
    }
    // a fresh contiguous window is planned by all lanes; the serial walk takes only what that leaves
    const bool windowed = fresh && st.round_stride == 16 && st.pending == 0 && !st.done && st.nstage < (uint32_t)SEG_GEN_MAX;
    if (windowed)
      plan_window(st, win4, nxt, list, reinterpret_cast<uint16_t*>(chunk), entry_of, stage, end, w, n_values, (uint32_t)SEG_GEN_MAX, tid);
    if (tid == 0 && !st.need_round && !st.done)
      plan_walk(st, win, stage, w, end, n_values, seg_groups_cap, (uint32_t)SEG_GEN_MAX, windowed ? 1u : ~0u);
    __syncthreads();
    if (st.status == PLAN_ST_GENERIC) fallback = true;
    const bool done = st.done != 0;
    const uint32_t vals = st.vals;
    uint32_t ns = st.nstage;
    // ---- cut chunks off the front of the stage while there is reason to
    while (!fallback && (done ? v0 < vals : (ns >= (uint32_t)SEG_GEN_MAX || vals - v0 >= span))) {
      uint32_t v1 = vals - v0 > span ? v0 + span : vals;
      if (!(done && v1 == vals)) v1 &= ~(uint32_t)(SEG_BLOCK - 1);
      if (v1 <= v0) {  // more than SEG_GEN_MAX runs inside one block of values: not for this path
        fallback = true;
        break;
      }
      // k = staged segments that start before v1 (the stage is sorted by start)
      uint32_t k = 0;
      {
        uint32_t lo = 0, hi = ns;  // start[lo] < v1 (segment 0 starts at or before v0), start[hi] >= v1
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if ((stage[mid].start4 >> 2) < v1)
            lo = mid;
          else
            hi = mid;
        }
        k = hi;
      }
      // short chunks over a dictionary that is swept go to k_dict_expand: they are taken from the pool's end
      const bool short_chunk = want_dict && v1 - v0 <= direct_max &&
                               (elem == 8 ? d.dict_len > DP_QUARTER_ENTRIES : (uint64_t)d.dict_len * 4u > 2ull * DS_SLICE_BYTES);
      // Chunks of dictionary pages that are nothing but whole bit-packed runs at one distance from
      // each other (the last one may be shorter) are marked regular, as the bulk path above marks its
      // chunks; a batch without any other chunk is gathered by the regular path of k_dict_pages_sliced.
      bool regular = false;
      uint32_t V = 1u << 20, S = 0;  // a single run: every unit lies in run 0
      if (want_dict) {
        const uint32_t s0 = stage[0].start4, c0 = stage[0].src;
        const uint32_t in_off = (uint32_t)(uintptr_t)d.data & 3u;
        bool ok = w >= 1 && k >= 1 && (s0 & 3u) == SEG_PACKED_BIT && ((s0 >> 2) & 3u) == 0 && in_off + d.data_len >= 12u &&
                  regular_dictionary;  // (that path keeps a dictionary resident or sweeps four quarters per plane)
        if (k >= 2) {
          V = (stage[1].start4 >> 2) - (s0 >> 2);
          S = stage[1].src - c0;
          // (the last run may be shorter than the others, not longer: a unit's run is its value index / V)
          ok = ok && V != 0 && (V & 3u) == 0 && V <= (1u << 20) && v1 - (stage[k - 1].start4 >> 2) <= V;
        }
        bool bad = false;
        for (uint32_t i = tid; i < k; i += PLAN_THREADS)
          bad = bad || stage[i].start4 != s0 + ((i * V) << 2) || stage[i].src != c0 + i * S;
        regular = ok && __builtin_amdgcn_ballot_w64(bad) == 0;
      }
      // (the short last chunk of a long regular page stays with the sliced gather: one more chunk of a sweep that
      // is running anyway costs less than a workgroup of its own in the left-over launch)
      const bool direct = short_chunk && (n_values <= direct_max || !regular);
      if (tid == 0) {
        // the first chunk of page p is pool chunk p (thousands of pages asking one counter for a slot
        // at the same time would queue up behind each other); further chunks come from the shared tail
        uint32_t slot = ~0u;
        if (page >= front_limit) {
          // no room for this page at all
        } else if (direct) {
          const uint32_t b = atomicAdd(cursor + PLAN_CTR_DIRECT, 1u);
          if (b < pool_chunks - front_limit) {
            slot = pool_chunks - 1u - b;
            if (emitted == 0) write_empty_chunk(pool, page);  // the sliced kernel finds nothing at the page's slot
          }
        } else if (emitted == 0) {
          slot = page;
        } else {
          const uint32_t f = atomicAdd(cursor + PLAN_CTR_TAIL, 1u);
          if (front_limit > n_pages && f < front_limit - n_pages) slot = n_pages + f;
        }
        st.chunk_slot = slot;
      }
      __syncthreads();
      const uint32_t slot = st.chunk_slot;
      if (slot == ~0u) {
        fallback = true;
        break;
      }
      for (uint32_t i = tid; i < k; i += PLAN_THREADS) chunk[i] = stage[i];
      if (tid == 0) {
        chunk[k].start4 = v1 << 2;  // sentinel
        chunk[k].src = 0;
        RleChunkHead* hd = reinterpret_cast<RleChunkHead*>(chunk + SEG_HEAD_SLOT);
        hd->data = (uint64_t)(uintptr_t)d.data;
        hd->out = (uint64_t)(uintptr_t)d.out;
        hd->v0 = v0;
        hd->v1 = v1;
        hd->nseg = k | nseg_tag;
        hd->w = w;
        hd->data_len = d.data_len;
        hd->stream_end = end;
        hd->page = page;
        hd->dict = (uint64_t)(uintptr_t)d.dict;
        hd->dict_len = d.dict_len;
        hd->reserved[0] = regular ? V : 0u;  // != 0: a chunk of equal runs
        hd->reserved[1] = regular ? S : 0u;
        if (regular) cursor[d.dict_len > 3u * DP_QUARTER_ENTRIES ? PLAN_CTR_SWEPT : PLAN_CTR_RESIDENT] = 1u;
        // A chunk that is not regular: the batch cannot take the regular path of k_dict_pages_sliced (flag in
        // counter 3), unless there are only a few such chunks among thousands (one run of eight equal ids
        // somewhere in a column): the first PLAN_IRREGULAR_LIST of them are listed, the regular path skips
        // them and the left-over launch expands them with a gather from memory.  (The flag is a store and
        // the list is short on purpose: thousands of atomics on one word queue up behind each other.)
        if (want_dict && !direct && !regular) {
          bool listed = false;
          if (regular_dictionary && direct_max != 0 &&
              __hip_atomic_load(cursor + PLAN_CTR_IRREGULAR_LISTED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < PLAN_IRREGULAR_LIST) {
            const uint32_t idx = atomicAdd(cursor + PLAN_CTR_IRREGULAR_LISTED, 1u);
            if (idx < PLAN_IRREGULAR_LIST) {
              cursor[PLAN_CTR_LIST0 + idx] = slot;
              listed = true;
            }
          }
          if (!listed) cursor[PLAN_CTR_IRREGULAR] = 1u;
        }
      }
      __syncthreads();
      uint8_t* table = reinterpret_cast<uint8_t*>(chunk + SEG_TABLE_SLOT);
      for (uint32_t b = tid; b < (uint32_t)SEG_TABLE_BLOCKS; b += PLAN_THREADS) {
        const uint32_t first = v0 + b * SEG_BLOCK;
        uint32_t lo = 0, hi = k;  // start[lo] <= first < start[hi]
        if (first >= v1) {
          lo = k - 1;
        } else {
          while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((chunk[mid].start4 >> 2) <= first)
              lo = mid;
            else
              hi = mid;
          }
        }
        table[b] = (uint8_t)lo;
      }
      __syncthreads();
      {
        const u32x4* src = reinterpret_cast<const u32x4*>(chunk);
        u32x4* dst = reinterpret_cast<u32x4*>(pool + (size_t)slot * SEG_CHUNK);
        for (uint32_t i = tid; i < (k + 2) / 2; i += PLAN_THREADS) dst[i] = src[i];                    // segments + sentinel
        for (uint32_t i = SEG_HEAD_SLOT / 2 + tid; i < (uint32_t)SEG_CHUNK / 2; i += PLAN_THREADS) dst[i] = src[i];  // header + table
      }
      // what stays staged: the segments from v1 on, led by the last one cut if it reaches past v1
      const uint32_t next_start = k < ns ? stage[k].start4 >> 2 : vals;
      const uint32_t j0 = next_start > v1 ? k - 1 : k;
      const uint32_t left = ns - j0;
      for (uint32_t base = 0; base < left; base += PLAN_THREADS) {
        RleSeg t = {0u, 0u};
        if (base + tid < left) t = stage[j0 + base + tid];
        __syncthreads();
        if (base + tid < left) stage[base + tid] = t;
        __syncthreads();
      }
      ns = left;
      v0 = v1;
      ++emitted;
    }
    __syncthreads();
    if (tid == 0) st.nstage = ns;
    __syncthreads();
    if (done || fallback) break;
  }
  if (tid == 0) {
    if (emitted == 0 && page < pool_chunks) write_empty_chunk(pool, page);
    plans[page].flags = fallback ? PLAN_F_FALLBACK : PLAN_F_DONE;
    if (fallback) cursor[PLAN_CTR_FALLBACK] = 1u;  // (a flag: the left-over launch then looks through the plan records)
    if (!fallback) {
      pqd_page_result r;
      r.status = st.status;
      r.values_written = st.vals;
      r.bytes_consumed = consumed_override != ~0u ? consumed_override : st.pos;
      r.reserved = 0;
      results[page] = r;
    }
  }
}

hipError_t launch_rle_plan_chunks(const pqd_page_desc* descs, pqd_page_result* results, void* plans, void* pool,
                                  uint32_t pool_chunks, uint32_t* cursor, uint32_t n_pages, bool want_dict, int elem,
                                  const DictPlanes& planes, bool try_small, hipStream_t stream, bool long_ok) {
  if (n_pages == 0) return hipSuccess;
  if (want_dict && elem == 8 && planes.n_slots != 0)
    hipLaunchKernelGGL(k_dict_slots, dim3((n_pages + 255u) / 256u), dim3(256), 0, stream, descs, n_pages, cursor, planes);
  const bool small = try_small && !want_dict && elem == 4;
  // PQD_DIRECT_MAX: chunks of dictionary pages of at most that many values are gathered from memory (0: none)
  const char* dm = getenv("PQD_DIRECT_MAX");
  const uint32_t direct_max = dm ? (uint32_t)atoi(dm) : DS_DIRECT_MAX;
  // PQD_WHOLE_MAX: dictionary pages with a stream of at most that many bytes, at most half of what their values would
  // take bit-packed, are decoded whole by the left-over launch (0: none)
  const char* wm = getenv("PQD_WHOLE_MAX");
  const uint32_t whole_max = wm ? (uint32_t)atoi(wm) : WP_MAX_STREAM;
  if (small) {  // small pages first: what it decodes the plan kernel skips
    hipError_t e = launch_rle_small_pages(descs, results, plans, n_pages, stream);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_rle_plan_chunks, dim3(n_pages), dim3(PLAN_THREADS), 0, stream, descs, results,
                     reinterpret_cast<RlePagePlan*>(plans), reinterpret_cast<RleSeg*>(pool), pool_chunks, cursor, n_pages,
                     want_dict ? 1 : 0, elem, planes, small ? 1 : 0, direct_max, whole_max, long_ok ? 1 : 0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || !long_ok) return e;
  return launch_whole_long_pages(descs, results, plans, cursor, n_pages, want_dict, elem, stream);
}

}  // namespace pqd
