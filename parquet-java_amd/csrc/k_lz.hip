// Page decompression on the device: raw Snappy and the LZ4 block format (and stored bytes: a copy).
//
// Replaces the step before initFromPage in ColumnChunkPageReadStore.readPage (parquet-hadoop/.../hadoop/
// ColumnChunkPageReadStore.java:151-181 V1, :232-262 V2: decompressor.decompress(bytes, uncompressed size)),
// i.e. SnappyDecompressor.uncompress -> org.xerial.snappy.Snappy.uncompress (codec/SnappyDecompressor.java:
// 27-35) and Lz4RawDecompressor.uncompress -> io.airlift.compress.lz4.Lz4Decompressor (codec/
// Lz4RawDecompressor.java:29-49).  Both formats are a byte-serial stream of "copy these literal bytes" and
// "repeat `length` bytes from `offset` back" elements: where an element starts is known only when the one before
// it has been read, and what a match copies may be what the match before it wrote.  A workgroup per page breaks both
// chains, 2 KiB of the stream (a window) at a time; each of the parse's steps is one wave's, the bytes of the output
// tiles are shared out among the executing waves.  Which waves those are, and whether the steps run one after the
// other or at once on successive windows, depends on the size of the call (see LW_PIPE16_PAGES below):
//   parse   (1) the window lies in LDS, a 32-byte segment per lane.  Every lane reads each place of its segment as
//               if a sequence (Snappy: one element; LZ4: literals + match) started there, last place first, and notes
//               where the chain from that place leaves the segment, how many elements it passes and how many bytes
//               they produce: a place's entry is that of the place its own sequence ends at.
//           (2) The wave follows the real chain through the window, one look-up per segment entered.
//           (3) Every lane that was entered walks its segment forward and writes its elements into a list: output
//               position and source (literal: place in the stream; match: offset).  A sequence that breaks a rule
//               (offset beyond the output, more output than the page holds, bytes beyond the stream) or needs
//               bytes that are not staged (a long LZ4 literal in front of its match) ends the window in front of
//               it; that one sequence is then taken by the wave-uniform code at the end of this file, which
//               reports a failure the way the formats' decoders do.
//   execute (4) The window's output is produced a tile of 4 KiB at a time, a lane per byte: the byte's element from
//               a bit mask of the element starts; a literal byte is fetched; a match byte becomes a pointer to its
//               source byte (for a match that overlaps itself: to the first period), or is fetched from memory
//               when the source lies in front of the tile.  Pointers are then followed by all bytes (p = p of p) until
//               every byte has its value; the waves do not wait for each other between rounds.  The tile leaves in
//               aligned 16-byte stores.
// Not HBM-bound: a page's time is the sum (or, with the steps on different waves, the longest) of its serial steps; with
// thousands of pages in flight the instructions per byte bound it (about twenty wave-instructions per 64 output bytes of
// repetitive text, half of them in the pointer rounds).
// Measured and dropped: an element per lane copying its bytes in order, with an LDS ring of the last 32 KiB of output
// (fewer instructions when matches reach far back, three times slower when most matches read what the 64 elements
// of their own batch write, as in columns of small numbers and repetitive text).
#include <cstdio>
#include <cstdlib>

#include "pqd_device.hpp"
#include "pqd_kernels.hpp"

namespace pqd {

#ifndef LW_FAR
#define LW_FAR 0
#endif
#if LW_FAR == 0
#define LW_FAR_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#elif LW_FAR == 1
#define LW_FAR_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#else
#define LW_FAR_LOAD(p) (*(p))
#endif
constexpr uint32_t LW_WIN = 2048;    // stream bytes per window
constexpr uint32_t LW_SLACK = 96;    // staged behind it: the headers of sequences that start in the window's last places
constexpr uint32_t LW_LENGTH_BYTES = 4;  // LZ4 length bytes read per length in the windows
constexpr uint32_t LW_HUGE = 1u << 18;  // elements from this length on are left to the wave-uniform code
constexpr uint32_t LW_TILE = 4096;   // output bytes per tile
constexpr uint32_t LW_ELEMS = 1408;  // elements per window at most (Snappy: 2 bytes each; LZ4: 2 per 3 bytes)
constexpr uint32_t LW_FAIL = 0x80000000u, LW_NONE = 0xFFFFFFFFu, LW_LIT = 0x80000000u;
constexpr uint16_t LW_HAS = 0x8000u;  // a tile cell that holds its byte
#ifndef LW_UNROLL
#define LW_UNROLL 4u
#endif
#ifndef LW_HOPS
#define LW_HOPS 2u  // pointers followed per round of a tile's resolution
#endif
// Kernel and waves per page by the size of the call.  A page's time is that of its serial chains (the parse: table,
// chain, elements; the steps of its tiles) rather than the CU's throughput, so few pages get many waves each:
//   at most 256 pages: k_lz_staged_pages with sixteen waves: the table's two halves, the chain and the elements by a
//                      wave each, twelve for the execution, all at once on five successive windows (95 KB of LDS: a
//                      page a CU);
//   at most 512 pages: the same with eight waves: table, chain, elements, five for the execution (77 KB: two pages a CU);
//   at most 768 pages: k_lz_pipe_pages, one wave parses while three execute the window before (52 KB: three pages a CU);
//   more:              k_lz_pages, four waves that parse, then execute (22.5 KB, 68 VGPRs: seven pages a CU).
// 1 MiB pages of text / small integers / doubles, Snappy: 256 pages 85 / 83 / 71 GB/s (k_lz_pages with 16 waves: 45 /
// 29 / 36), 512 pages 100 / 107 / 102, 768 pages 110 / 91 / 117, 1792 pages 160 / 130 / 153; LZ4: 73 / 90 / 63,
// 96 / 100 / 96, 112 / 90 / 108, 147 / 137 / 139.
constexpr uint32_t LW_PIPE16_PAGES = 256, LW_PIPE8_PAGES = 512, LW_PIPE4_PAGES = 768;

__device__ __forceinline__ uint32_t lw_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// window byte q in LDS (no padding between the lanes' segments: the instructions of an address with padding cost more
// than the bank conflicts of one without)
__device__ __forceinline__ uint32_t lw_at(uint32_t q) { return q; }

// What the parse of a window tells its execution.
struct LwWindow {
  uint32_t w0, staged;    // the stream bytes in LDS: [w0, staged)
  uint32_t o0, out_end;   // the output of the window's elements: [o0, out_end)
  uint32_t n_elems, new_ip;
  uint32_t stopped;       // the chain ended in front of a sequence that is taken by itself
  uint32_t win_slot;      // where the window's bytes lie
};

struct LwElems {
  uint32_t e_out[LW_ELEMS + 8];  // output position of element i; [n] = where the window's output ends
  uint32_t e_src[LW_ELEMS + 8];  // LW_LIT | stream position of a literal's first byte, or a match's offset
};

struct LwTiles {
  uint32_t t_lo[64], t_hi[64];  // per 64 bytes of the tile: the bytes at which an element starts
  uint32_t t_before[64];        // the element that holds the byte in front of those 64
};

// One window at a time, parsed and then executed by the same waves (22.5 KB: seven pages a CU; the table of places is
// done with when the list of elements is written).
struct LwLds {
  __attribute__((aligned(16))) uint8_t win_[LW_WIN + LW_SLACK + 32];
  union {
    // parse: per place .x = where the chain from it leaves the segment (LW_FAIL: it stops inside),
    // .y = elements on the way << 24 | output bytes of the sequences on the way
    u32x2 place[64 * 33];
    struct {  // the elements, and the tile they are executed in
      LwElems e;
      __attribute__((aligned(16))) uint16_t cell[LW_TILE];
    } x;
  } u;
  LwTiles t;
  LwWindow w_;
  uint32_t after_ip, after_op, after_failed;  // where a sequence taken by itself ended
  // (a tile's first 512 bytes, before its cells are written)
  __device__ __forceinline__ uint32_t* bits() { return reinterpret_cast<uint32_t*>(u.x.cell); }
  __device__ __forceinline__ uint8_t* win(uint32_t) { return win_; }
  __device__ __forceinline__ u32x2* place() { return u.place; }
  __device__ __forceinline__ LwElems& elems(uint32_t) { return u.x.e; }
  __device__ __forceinline__ uint16_t* cell() { return u.x.cell; }
  __device__ __forceinline__ LwWindow& window(uint32_t) { return w_; }
};

// Two windows: the first wave parses one while the others execute the one before it (50 KB: three pages a CU).
struct LwPipeLds {
  __attribute__((aligned(16))) uint8_t win_[2][LW_WIN + LW_SLACK + 32];
  u32x2 place_[64 * 33];
  LwElems e_[2];
  __attribute__((aligned(16))) uint16_t cell_[LW_TILE];
  LwTiles t;
  // (a tile's first 512 bytes, before its cells are written)
  __device__ __forceinline__ uint32_t* bits() { return reinterpret_cast<uint32_t*>(cell_); }
  LwWindow w_[2];
  uint32_t after_ip, after_op, after_failed;
  uint32_t arrived, abort;  // the executing waves' barrier: arrivals so far; a wait that ran out of patience
  __device__ __forceinline__ uint8_t* win(uint32_t s) { return win_[s]; }
  __device__ __forceinline__ u32x2* place() { return place_; }
  __device__ __forceinline__ LwElems& elems(uint32_t s) { return e_[s]; }
  __device__ __forceinline__ uint16_t* cell() { return cell_; }
  __device__ __forceinline__ LwWindow& window(uint32_t s) { return w_[s]; }
};

// What the chaining wave hands to the one that lists the elements.
struct LwChainRecord {
  uint32_t entry[64], first_elem[64], first_out[64];
  uint32_t w0, win_slot, ip, op;  // the window, and where the chain entered it
  uint32_t next_ip, next_op;      // where the chain left it
};

// Four windows in flight, a wave for each of the parse's three serial steps: the first makes the table of places of a
// window (windows then start at multiples of LW_WIN, so that a table does not depend on where the chain enters it),
// the second follows the chain through the window before it, the third lists the elements of the one before that, the
// others execute the one before that (76 KB: two pages a CU).  SPLIT: five windows, the table's upper and lower half
// by a wave each, one after the other (95 KB: for the workgroups of sixteen waves, which have a CU to themselves).
template <bool SPLIT>
struct LwStagedLdsT {
  static constexpr uint32_t WIN_SLOTS = SPLIT ? 5u : 4u, PLACE_SLOTS = SPLIT ? 3u : 2u;
  __attribute__((aligned(16))) uint8_t win_[WIN_SLOTS][LW_WIN + LW_SLACK + 32];
  u32x2 place_[PLACE_SLOTS][64 * 33];
  LwChainRecord k_[2];
  LwElems e_[2];
  __attribute__((aligned(16))) uint16_t cell_[LW_TILE];
  LwTiles t;
  LwWindow w_[2];
  // the masks of a listed window's first tiles, made by the wave that lists the elements (two tiles without SPLIT: what
  // LDS is left with two pages a CU)
  static constexpr uint32_t READY_TILES = SPLIT ? 4u : 2u;
  LwTiles ready_[2][READY_TILES];
  uint32_t n_ready[2];
  uint32_t parse_bits[128];
  uint32_t after_ip, after_op, after_failed;
  uint32_t arrived, abort;
  // (a tile's first 512 bytes, before its cells are written)
  __device__ __forceinline__ uint32_t* bits() { return reinterpret_cast<uint32_t*>(cell_); }
  __device__ __forceinline__ LwTiles* ready(uint32_t s) { return ready_[s]; }
  __device__ __forceinline__ uint8_t* win(uint32_t s) { return win_[s]; }
  __device__ __forceinline__ u32x2* place(uint32_t s) { return place_[s]; }
  __device__ __forceinline__ LwChainRecord& chain(uint32_t s) { return k_[s]; }
  __device__ __forceinline__ LwElems& elems(uint32_t s) { return e_[s]; }
  __device__ __forceinline__ uint16_t* cell() { return cell_; }
  __device__ __forceinline__ LwWindow& window(uint32_t s) { return w_[s]; }
};

// The barrier between the steps of a tile.  All waves of the workgroup execute: the hardware's.
struct LwAllWaves {
  __device__ __forceinline__ void sync() { __syncthreads(); }
  __device__ __forceinline__ bool aborted() const { return false; }
  __device__ __forceinline__ void give_up() {}
};
// All waves but the parsing one execute: every wave adds one to a counter in LDS and waits until all have (the counter
// only grows; the comparison survives its wrap).  The waves of a workgroup are resident together, so the wait ends; a
// wait of more than LW_PATIENCE looks nevertheless gives the page up (PQD_DECODE) rather than never ending.
constexpr uint32_t LW_PATIENCE = 1u << 24;
struct LwExecWaves {
  uint32_t* arrived;
  uint32_t* abort;
  uint32_t target, n_waves, lane;
  __device__ __forceinline__ void sync() {
    target += n_waves;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    uint32_t looks = 0;
    while ((int32_t)(__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - target) < 0) {
      if (++looks > LW_PATIENCE || __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
        __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  __device__ __forceinline__ bool aborted() const { return __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u; }
  __device__ __forceinline__ void give_up() { __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
};

#ifdef LW_PROF
#define LW_STAMP(k) { const uint64_t now_ = __builtin_readcyclecounter(); prof[k] += now_ - prof_t; prof_t = now_; }
#define LW_TICK prof_t = __builtin_readcyclecounter();
#define LW_PROF_ARGS , uint64_t* prof, uint64_t& prof_t
#define LW_PROF_PASS , prof, prof_t
#else
#define LW_STAMP(k)
#define LW_TICK
#define LW_PROF_ARGS
#define LW_PROF_PASS
#endif

// (development build -DLW_WATCHDOG: a loop that runs four million times says so and ends)
#ifdef LW_WATCHDOG
#define LW_DOG(var, where)                                                                        \
  if (++(var) > (1u << 22)) {                                                                     \
    if ((threadIdx.x & 63u) == 0) printf("k_lz watchdog: loop %d, block %u wave %u\n", where, blockIdx.x, threadIdx.x >> 6); \
    break;                                                                                        \
  }
#else
#define LW_DOG(var, where) (void)(var);
#endif

struct LwSeq {
  bool ok;
  uint32_t next, lit_len, lit_pos, mlen, offset;
};

// The sequence that would start at stream position p.  Bytes are read from the staged window (positions below
// `staged`); ok = false: the sequence breaks a rule of the format or needs bytes that are not staged.
template <int CODEC>
__device__ __forceinline__ LwSeq lw_parse(const uint8_t* win, uint32_t w0, uint32_t p, uint32_t in_end, uint32_t staged) {
  LwSeq s = {false, 0u, 0u, 0u, 0u, 0u};
  if (p >= in_end) return s;
#define LW_B(q) ((uint32_t)win[lw_at((q) - w0)])
  if (CODEC == PQD_CODEC_SNAPPY) {
    // (without branches: the lanes of a wave meet all kinds of elements at once)
    const uint32_t tag = LW_B(p), kind = tag & 3u, n6 = tag >> 2;
    const uint32_t b1 = LW_B(p + 1), b2 = LW_B(p + 2), b3 = LW_B(p + 3), b4 = LW_B(p + 4);
    const uint32_t rest = b1 | (b2 << 8) | (b3 << 16) | (b4 << 24);
    const bool lit = kind == 0u;
    // as a literal: lengths above 60 in 1..4 bytes behind the tag
    const uint32_t nb = n6 >= 60u ? n6 - 59u : 0u;
    const uint32_t field = rest & (uint32_t)((1ull << (8u * nb)) - 1ull);
    const bool field_fine = field < in_end;  // (longer than the stream otherwise; no overflow below)
    const uint32_t lit_len = (nb ? (field_fine ? field : 0u) : n6) + 1u;
    // as a copy
    const uint32_t copy_hdr = kind == 1u ? 2u : kind == 2u ? 3u : 5u;
    const uint32_t copy_len = kind == 1u ? 4u + (n6 & 7u) : 1u + n6;
    const uint32_t offset = kind == 1u ? ((tag >> 5) << 8) | b1 : kind == 2u ? rest & 0xFFFFu : rest;
    s.lit_len = lit ? lit_len : 0u;
    s.lit_pos = p + 1u + nb;
    s.mlen = lit ? 0u : copy_len;
    s.offset = offset;
    s.next = p + (lit ? 1u + nb + lit_len : copy_hdr);
    s.ok = s.next <= in_end && (lit ? field_fine : offset != 0u);
  } else {
    const uint32_t token = LW_B(p);
    uint32_t q = p + 1u, lit = token >> 4;
    if (lit == 15u) {
      // (a length of more than LW_LENGTH_BYTES bytes: a sequence for the wave-uniform code, which reads them 64 at a
      // time; a run of 255s in the stream would otherwise be walked from every place in it)
      uint32_t b, k = 0;
      do {
        if (q >= in_end || q >= staged || ++k > LW_LENGTH_BYTES) return s;
        b = LW_B(q);
        ++q;
        lit += b;
      } while (b == 255u);
    }
    if (lit > in_end - q) return s;
    s.lit_len = lit;
    s.lit_pos = q;
    q += lit;
    if (q == in_end) {  // the last sequence has no match
      s.next = q;
      s.ok = true;
      return s;
    }
    if (in_end - q < 2u || q + 2u > staged) return s;
    s.offset = LW_B(q) | (LW_B(q + 1) << 8);
    q += 2u;
    uint32_t m = token & 15u;
    if (m == 15u) {
      uint32_t b, k = 0;
      do {
        if (q >= in_end || q >= staged || ++k > LW_LENGTH_BYTES) return s;
        b = LW_B(q);
        ++q;
        m += b;
      } while (b == 255u);
    }
    if (s.offset == 0u) return s;
    s.mlen = m + 4u;
    s.next = q;
    s.ok = true;
  }
#undef LW_B
  return s;
}

// stream bytes [w0, staged) -> LDS; thread k of the n that share the work passes first = 16 k, stride = 16 n
__device__ __forceinline__ void lw_stage(uint8_t* win, const PQD_GLOBAL uint8_t* src, uint32_t w0, uint32_t staged, uint32_t first, uint32_t stride) {
  for (uint32_t t = first; t < staged - w0; t += stride) {
    if (t + 16u <= staged - w0) {
      u32x4 v;
      __builtin_memcpy(&v, (const uint8_t*)(src + w0 + t), 16);
      *reinterpret_cast<u32x4*>(win + t) = v;
    } else {
      for (uint32_t k = 0; t + k < staged - w0; ++k) win[lw_at(t + k)] = src[w0 + t + k];
    }
  }
}

__device__ __forceinline__ uint32_t lw_staged_end(uint32_t w0, uint32_t in_end) {
  return in_end - w0 < LW_WIN + LW_SLACK ? in_end : w0 + LW_WIN + LW_SLACK;
}

// The table of places of the window that starts at stream position w0 (its bytes lie in `win`), by one wave:
// the places of this lane's segment, last to first, two at a time (a sequence is two bytes or more, so neither of the
// two continues at the other; the one exception is dealt with below).
// (first / last: the places first, first - 1 .. last, last - 1 of every segment, for two waves that make a table's
// upper and lower half one after the other)
template <int CODEC>
__device__ __forceinline__ void lw_places(const uint8_t* win, u32x2* place, uint32_t w0, uint32_t in_end, uint32_t lane, int first = 31,
                                          int last = 1) {
  const uint32_t staged = lw_staged_end(w0, in_end);
  const uint32_t seg = w0 + lane * 32u;
  for (int pl = first; pl >= last; pl -= 2) {
    LwSeq s[2];
    uint32_t j[2];
    u32x2 at_j[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      s[u] = lw_parse<CODEC>(win, w0, seg + (uint32_t)(pl - u), in_end, staged);
      s[u].ok = s[u].ok && s[u].lit_len < LW_HUGE && s[u].mlen < LW_HUGE;  // (a segment's sum stays below 2^24)
      // a sequence that ends inside the segment continues with the entry of the place it ends at (the look-up is made
      // in any case, at a place inside the segment)
      j[u] = s[u].ok && s[u].next < seg + 32u ? s[u].next - seg : 31u;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) at_j[u] = place[lane * 33u + j[u]];
    u32x2 e[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bool inside = s[u].ok && s[u].next < seg + 32u;
      // (LZ4's last sequence may be its token alone: the lower place of the pair then continues at the upper one, whose
      // entry is not in the table yet)
      if (u == 1 && j[1] == (uint32_t)pl) at_j[1] = e[0];
      e[u].x = !s[u].ok ? LW_FAIL : inside ? at_j[u].x : s[u].next;
      e[u].y = !s[u].ok ? 0u : s[u].lit_len + s[u].mlen + (((s[u].lit_len ? 1u : 0u) + (s[u].mlen ? 1u : 0u)) << 24) + (inside ? at_j[u].y : 0u);
      place[lane * 33u + (uint32_t)(pl - u)] = e[u];
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// What the chain through a window leaves for a lane: where it enters the lane's segment (LW_NONE: it does not), and
// how many elements and output bytes lie in front of that place.
struct LwChained {
  uint32_t entry, first_elem, first_out;
};

// The chain of sequences through a window whose table of places is made, by one wave: from stream position ip on
// (output position op; ip lies in the window) as far as the window, or the chain of well-formed sequences, goes.
// next_ip / next_op: where it leaves the window (what counts is where the listing of the elements ends: the two agree
// unless that stops in front of a sequence, or the stream ends in the window).
__device__ __forceinline__ LwChained lw_chain(const u32x2* place, uint32_t w0, uint32_t ip, uint32_t op, uint32_t in_end, uint32_t cap,
                                              uint32_t lane, uint32_t& next_ip, uint32_t& next_op) {
  LwChained c = {LW_NONE, 0u, 0u};
  uint32_t at = ip, elems = 0;
  uint64_t made = op;
  uint32_t dog1 = 0;
  while (at < w0 + LW_WIN && at < in_end && made <= cap) {
    LW_DOG(dog1, 1)
    const uint32_t k = (at - w0) >> 5, pl = (at - w0) & 31u;
    const u32x2 e = place[k * 33u + pl];
    const uint32_t ex = lw_uni(e.x), m = lw_uni(e.y);
    if (lane == k) {
      c.entry = at;
      c.first_elem = elems;
      c.first_out = (uint32_t)made;  // (at most cap)
    }
    elems += m >> 24;
    made += m & 0xFFFFFFu;
    if (ex & LW_FAIL) break;
    at = ex;
  }
  next_ip = at;
  next_op = (uint32_t)made;
  return c;
}

// The elements of a chained window, by one wave: every lane that the chain entered walks its segment forward and
// writes its elements into the list E; the record *out says where the list ends.  A sequence that breaks a rule or
// needs bytes that are not staged ends the list in front of it (out->stopped).
template <int CODEC>
__device__ __forceinline__ void lw_elements(const uint8_t* win, LwChained c, LwElems& E, LwWindow* out, uint32_t win_slot, uint32_t w0,
                                            uint32_t ip, uint32_t op, uint32_t in_end, uint32_t cap, uint32_t lane) {
  const uint32_t staged = lw_staged_end(w0, in_end);
  const uint32_t seg = w0 + lane * 32u;
  uint32_t entry = c.entry;
  uint32_t mine = 0, end = 0;
  uint64_t out_end = 0;
  bool stop = false;
  if (entry != LW_NONE) {
    uint32_t q = entry, i = c.first_elem;
    uint64_t o = c.first_out;
    uint32_t dog2 = 0;
    while (q < seg + 32u && q < in_end) {
      LW_DOG(dog2, 2)
      const LwSeq s = lw_parse<CODEC>(win, w0, q, in_end, staged);
      if (!s.ok || s.lit_len >= LW_HUGE || s.mlen >= LW_HUGE || (s.mlen && s.offset > o + s.lit_len) || o + s.lit_len + s.mlen > cap) {
        stop = true;
        break;
      }
      if (s.lit_len) {
        E.e_out[i] = (uint32_t)o;
        E.e_src[i] = LW_LIT | s.lit_pos;
        o += s.lit_len;
        ++i;
      }
      if (s.mlen) {
        E.e_out[i] = (uint32_t)o;
        E.e_src[i] = s.offset;
        o += s.mlen;
        ++i;
      }
      q = s.next;
    }
    mine = i - c.first_elem;
    end = q;
    out_end = o;
  }
  // lanes behind one that stopped were entered for nothing
  const uint64_t stops = __builtin_amdgcn_ballot_w64(stop);
  if (stops && lane > (uint32_t)__builtin_ctzll(stops)) {
    entry = LW_NONE;
    mine = 0;
  }
  const uint64_t entered = __builtin_amdgcn_ballot_w64(entry != LW_NONE);
  const uint32_t n = lw_uni((uint32_t)__builtin_amdgcn_readlane((int)wave_scan_inclusive(mine), 63));
  const uint32_t last = entered ? lw_uni(63u - (uint32_t)__builtin_clzll(entered)) : 0u;
  const uint32_t stream_end = (uint32_t)__builtin_amdgcn_readlane((int)end, (int)last);
  const uint32_t output_end = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)out_end, (int)last);  // (at most cap)
  if (lane == 0) {
    E.e_out[entered ? n : 0u] = entered ? output_end : op;
    // (not entered cannot be: ip lies in the window)
    const LwWindow w = {w0, staged, op, entered ? output_end : op, entered ? n : 0u, entered ? stream_end : ip, (!entered || stops != 0) ? 1u : 0u, win_slot};
    *out = w;
  }
}

// Both by the same wave.
template <int CODEC>
__device__ __forceinline__ void lw_chain_elements(const uint8_t* win, const u32x2* place, LwElems& E, LwWindow* out, uint32_t win_slot,
                                                  uint32_t w0, uint32_t ip, uint32_t op, uint32_t in_end, uint32_t cap,
                                                  uint32_t lane LW_PROF_ARGS) {
  uint32_t next_ip, next_op;
  const LwChained c = lw_chain(place, w0, ip, op, in_end, cap, lane, next_ip, next_op);
  asm volatile("" ::: "memory");  // (where the list takes the table's room: no look-up moves behind this line)
  __builtin_amdgcn_wave_barrier();
  LW_STAMP(2)
  lw_elements<CODEC>(win, c, E, out, win_slot, w0, ip, op, in_end, cap, lane);
  LW_STAMP(3)
}

// The parse of the window that starts at ip, by one wave: table, chain and elements.  The window's bytes lie in L.win(slot).
template <int CODEC, class LDS>
__device__ __forceinline__ void lw_parse_window(LDS& L, uint32_t slot, uint32_t ip, uint32_t op, uint32_t in_end, uint32_t cap,
                                                uint32_t lane LW_PROF_ARGS) {
  lw_places<CODEC>(L.win(slot), L.place(), ip, in_end, lane);
  LW_STAMP(1)
  lw_chain_elements<CODEC>(L.win(slot), L.place(), L.elems(slot), &L.window(slot), slot, ip, ip, op, in_end, cap, lane LW_PROF_PASS);
}

// steps of 64 bytes that one of nw executing waves takes at a time: the steps of a tile are shared out evenly
__host__ __device__ constexpr uint32_t lw_steps_at_a_time(uint32_t nw) { return nw >= 16u ? 4u : nw >= 12u ? 5u : LW_UNROLL; }
// a tile is as many steps (of 64 at most) as nw waves take in whole turns
__host__ __device__ constexpr uint32_t lw_tile_bytes(uint32_t nw) {
  return nw * lw_steps_at_a_time(nw) >= 64u ? LW_TILE : (64u / (nw * lw_steps_at_a_time(nw))) * (nw * lw_steps_at_a_time(nw)) * 64u;
}

// The element starts inside the tile (tb, tb + tile) of a window's output as bit masks, 64 bytes of the tile per lane,
// by one wave.  e_lo: elements below it start at or in front of the tile (kept from tile to tile; 0 is always right).
// bits: 128 words of LDS for this wave alone.
__device__ __forceinline__ void lw_tile_masks(const LwElems& E, uint32_t n_elems, uint32_t o1, uint32_t tb, uint32_t tile, uint32_t& e_lo,
                                              uint32_t* bits, LwTiles& T, uint32_t lane) {
  const uint32_t te = o1 - tb < tile ? o1 : tb + tile;
  bits[lane] = 0;
  bits[lane + 64u] = 0;
  __builtin_amdgcn_wave_barrier();
  uint32_t dog3 = 0;
  for (;;) {
    LW_DOG(dog3, 3)
    const uint32_t i = e_lo + lane;
    const bool in_front = i < n_elems && E.e_out[i] <= tb;
    const uint64_t m = __builtin_amdgcn_ballot_w64(in_front);
    e_lo += (uint32_t)__builtin_popcountll(m);
    if (m != ~0ull) break;
  }
  uint32_t dog4 = 0;
  for (uint32_t b = e_lo;; b += 64u) {
    LW_DOG(dog4, 4)
    const uint32_t i = b + lane;
    const uint32_t eo = i < n_elems ? E.e_out[i] : 0xFFFFFFFFu;
    const bool inside = eo < te;
    if (inside) atomicOr(&bits[(eo - tb) >> 5], 1u << ((eo - tb) & 31u));
    if (__builtin_amdgcn_ballot_w64(inside) != ~0ull) break;
  }
  __builtin_amdgcn_wave_barrier();
  const uint32_t lo = bits[2u * lane], hi = bits[2u * lane + 1u];
  const uint32_t in_seg = (uint32_t)__builtin_popcount(lo) + (uint32_t)__builtin_popcount(hi);
  T.t_lo[lane] = lo;
  T.t_hi[lane] = hi;
  T.t_before[lane] = e_lo - 1u + wave_scan_inclusive(in_seg) - in_seg;  // the element of the byte in front of the segment
  __builtin_amdgcn_wave_barrier();
}

// The output of a parsed window, tile by tile, by XWAVES waves (this one is number xw of them); bar.sync() is their
// barrier.  The tiles are in memory when it returns.  ready[0 .. n_ready): the masks of the window's first tiles, where
// the parse has made them already.
template <uint32_t XWAVES, class LDS, class BAR>
__device__ __forceinline__ void lw_execute(LDS& L, uint32_t slot, const PQD_GLOBAL uint8_t* src, PQD_GLOBAL uint8_t* dst, uint32_t xw,
                                           uint32_t lane, BAR& bar, const LwTiles* ready, uint32_t n_ready LW_PROF_ARGS) {
  constexpr uint32_t NW = XWAVES;  // waves that share the bytes of a tile
  constexpr uint32_t UN = lw_steps_at_a_time(NW);
  constexpr uint32_t TILE = lw_tile_bytes(NW);
  const LwWindow W = L.window(slot);
  const uint8_t* win = L.win(W.win_slot);
  const LwElems& E = L.elems(slot);
  const uint32_t w0 = W.w0, staged = W.staged, n_elems = W.n_elems, o0 = W.o0, o1 = W.out_end;
  const uint32_t xtid = xw * 64u + lane;
  uint16_t* cell = L.cell();
  uint32_t* start_bits = L.bits();
  uint32_t e_lo = 0;  // (the first wave's: elements below it start at or in front of the tile)
  uint32_t ti = 0;
  for (uint32_t tb = o0; tb < o1; tb += TILE, ++ti) {
    const uint32_t te = o1 - tb < TILE ? o1 : tb + TILE;
    const bool have = ti < n_ready;
    const LwTiles& T = have ? ready[ti] : L.t;
    if (!have) {
      if (xw == 0) lw_tile_masks(E, n_elems, o1, tb, TILE, e_lo, start_bits, L.t, lane);
      bar.sync();
    }
    LW_STAMP(4)
    const uint32_t a = (uint32_t)((uintptr_t)(dst + tb) & 15u);
    const uint64_t upto = (2ull << lane) - 1ull;
    // a word per byte of the tile: LW_HAS | the byte, or the tile position of the byte it repeats
    const uint32_t n_steps = (te - tb + 63u) >> 6;
    bool open = false;
    for (uint32_t t0 = xw * UN; t0 < n_steps; t0 += NW * UN) {
      uint32_t eo[UN], es[UN];
      uint16_t word[UN];
#pragma unroll
      for (uint32_t u = 0; u < UN; ++u) {
        const uint32_t t = t0 + u < n_steps ? t0 + u : n_steps - 1u;
        const uint64_t st = ((uint64_t)T.t_hi[t] << 32) | T.t_lo[t];
        uint32_t i = T.t_before[t] + (uint32_t)__builtin_popcountll(st & upto);
        i = i < n_elems ? i : n_elems - 1u;  // (bytes behind the output's end: any element, the result is not used)
        eo[u] = E.e_out[i];
        es[u] = E.e_src[i];
      }
#pragma unroll
      for (uint32_t u = 0; u < UN; ++u) {
        const uint32_t o = tb + (t0 + u) * 64u + lane;
        const uint32_t rel = o - eo[u];
        word[u] = LW_HAS;
        if (t0 + u < n_steps && o < te) {
          if (es[u] & LW_LIT) {
            const uint32_t at = (es[u] & ~LW_LIT) + rel;
            word[u] = LW_HAS | (uint16_t)(at < staged ? win[lw_at(at - w0)] : src[at]);
          } else {
            uint32_t from = o - es[u];
            if (rel >= es[u]) from = eo[u] - es[u] + rel % es[u];  // a match that repeats itself: the first period
            if (from < tb) {
              word[u] = LW_HAS | (uint16_t)LW_FAR_LOAD(dst + from);  // (written by this workgroup and waited for: read where the stores went)
            } else {
              word[u] = (uint16_t)(from - tb);
              open = true;
            }
          }
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < UN; ++u)
        if (t0 + u < n_steps) cell[(t0 + u) * 64u + lane] = word[u];
    }
    bar.sync();
    LW_STAMP(5)
    // Every byte follows its pointer until it arrives at a byte that has its value.  The waves do not wait for each
    // other between rounds: whatever a cell holds at any time stands for its byte (a pointer to an equal byte further
    // to the front, or the value), and pointers only lead to the front, where the cells get done first.
    if (__builtin_amdgcn_ballot_w64(open)) {
      uint32_t dog5 = 0, rounds = 0;
      for (;;) {
        LW_DOG(dog5, 5)
        asm volatile("" ::: "memory");  // (the cells are read again: other waves write them)
        open = false;
        for (uint32_t t0 = xw * UN; t0 < n_steps; t0 += NW * UN) {
          uint16_t mine_[UN], other[UN];
#pragma unroll
          for (uint32_t u = 0; u < UN; ++u) mine_[u] = t0 + u < n_steps ? cell[(t0 + u) * 64u + lane] : LW_HAS;
#pragma unroll
          for (uint32_t u = 0; u < UN; ++u) other[u] = (mine_[u] & LW_HAS) ? LW_HAS : cell[mine_[u]];
          // (LW_HOPS pointers a round: chains inside a tile are short, and a look-up costs less than a round)
#pragma unroll
          for (uint32_t h = 1; h < LW_HOPS; ++h) {
#pragma unroll
            for (uint32_t u = 0; u < UN; ++u)
              if (!(other[u] & LW_HAS)) other[u] = cell[other[u]];
          }
#pragma unroll
          for (uint32_t u = 0; u < UN; ++u) {
            if (!(mine_[u] & LW_HAS)) {
              cell[(t0 + u) * 64u + lane] = other[u];  // its value, or the byte that one repeats
              open = open || !(other[u] & LW_HAS);
            }
          }
        }
#ifdef LW_COUNT_ROUNDS
        prof[7] += 1000;
#endif
        if (!__builtin_amdgcn_ballot_w64(open) || bar.aborted()) break;
        if (++rounds > LW_PATIENCE) {  // (cannot be: every round brings the frontmost open cell to its value)
          bar.give_up();
          break;
        }
      }
    }
    bar.sync();
    LW_STAMP(6)
    // tile -> memory: whole aligned 16-byte pieces, the ragged ends byte by byte
    {
      PQD_GLOBAL uint8_t* gbase = dst + tb - a;  // 16-byte aligned
      const uint32_t stop_at = a + (te - tb);
      for (uint32_t t = xtid * 16u; t < stop_at; t += NW * 1024u) {
        // the bytes of tile positions t - a .. t - a + 15
        uint32_t packed[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) {
          const uint32_t q = t + k - a;  // (wraps below a: not used then)
          const uint32_t v = (t + k >= a && t + k < stop_at) ? cell[q] & 0xFFu : 0u;
          packed[k >> 2] |= v << (8u * (k & 3u));
        }
        if (t >= a && t + 16u <= stop_at) {
          const u32x4 v = {packed[0], packed[1], packed[2], packed[3]};
          *(PQD_GLOBAL u32x4*)(gbase + t) = v;
        } else {
          for (uint32_t k = 0; k < 16u; ++k)
            if (t + k >= a && t + k < stop_at) gbase[t + k] = (uint8_t)(packed[k >> 2] >> (8u * (k & 3u)));
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tile is in memory when the next one reads from it
    bar.sync();
#ifndef LW_COUNT_ROUNDS
    LW_STAMP(7)
#endif
  }
}

// One window, parsed by the first wave and then executed by all WAVES waves of the workgroup: the sequences from
// stream position ip on; ip / op move behind them.  Returns true when the chain stopped in front of a sequence that
// the caller has to take by itself.
template <int CODEC, uint32_t WAVES>
__device__ __forceinline__ bool lw_window(LwLds& L, const PQD_GLOBAL uint8_t* src, PQD_GLOBAL uint8_t* dst, uint32_t& ip, uint32_t& op,
                                          uint32_t in_end, uint32_t cap, uint32_t tid LW_PROF_ARGS) {
  const uint32_t lane = tid & 63u, wave = tid >> 6;
  __syncthreads();
  lw_stage(L.win(0), src, ip, lw_staged_end(ip, in_end), tid * 16u, WAVES * 1024u);
  __syncthreads();
  LW_STAMP(0)
  if (wave == 0) lw_parse_window<CODEC>(L, 0u, ip, op, in_end, cap, lane LW_PROF_PASS);
  __syncthreads();
  LwAllWaves bar;
  lw_execute<WAVES>(L, 0u, src, dst, wave, lane, bar, nullptr, 0u LW_PROF_PASS);
  const LwWindow W = L.window(0);  // (rewritten behind the next window's first barrier)
  op = W.out_end;
  ip = W.new_ip;
  return W.stopped != 0u;
}

// ---- one sequence, wave-uniformly, with its bytes moved from memory to memory: the sequences the windows stop at
__device__ __forceinline__ uint32_t lw_byte(const PQD_GLOBAL uint8_t* src, uint32_t pos) { return lw_uni(src[pos]); }

__device__ __forceinline__ void lw_copy_literal(const PQD_GLOBAL uint8_t* from, PQD_GLOBAL uint8_t* to, uint32_t len, uint32_t lane) {
  const uint32_t pieces = len >> 4;
  uint32_t i = lane;
  for (; i + 7u * 64u < pieces; i += 8u * 64u) {  // eight loads on their way before the first store
    u32x4 v[8];
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k) __builtin_memcpy(&v[k], (const uint8_t*)(from + (i + k * 64u) * 16u), 16);
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k) __builtin_memcpy((uint8_t*)(to + (i + k * 64u) * 16u), &v[k], 16);
  }
  for (; i < pieces; i += 64u) {
    u32x4 v;
    __builtin_memcpy(&v, (const uint8_t*)(from + i * 16u), 16);
    __builtin_memcpy((uint8_t*)(to + i * 16u), &v, 16);
  }
  for (uint32_t j = (pieces << 4) + lane; j < len; j += 64u) to[j] = from[j];
}

// `len` bytes from `offset` back (memory to memory; the bytes were stored by this workgroup and are waited for).
// A match that reads its own output repeats the `offset` bytes in front of it: for offsets below 64 the lanes hold
// that pattern and hand it round; longer ones are copied a period at a time.
__device__ __forceinline__ void lw_copy_match(PQD_GLOBAL uint8_t* dst, uint32_t op, uint32_t offset, uint32_t len, uint32_t lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (offset < 64u) {
    uint32_t ph = lane;  // lane mod offset: six conditional subtractions
    uint32_t step = 64u;  // 64 mod offset
#pragma unroll
    for (int k = 5; k >= 0; --k) {
      const uint32_t t = offset << k;
      ph = ph >= t ? ph - t : ph;
      step = step >= t ? step - t : step;
    }
    step = step >= offset ? step - offset : step;
    const uint32_t pattern = lane < offset ? (uint32_t)LW_FAR_LOAD(dst + (op - offset + lane)) : 0u;
    for (uint32_t b = 0; b < len; b += 64u) {
      const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ph << 2), (int)pattern);
      if (b + lane < len) dst[op + b + lane] = (uint8_t)v;
      ph += step;
      ph = ph >= offset ? ph - offset : ph;
    }
    return;
  }
  // pieces of at most `offset` bytes: a piece reads what lies in front of it
  for (uint32_t done = 0; done < len;) {
    const uint32_t piece = len - done < offset ? len - done : offset;
    if (done) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (uint32_t i = lane; i < piece; i += 64u) dst[op + done + i] = LW_FAR_LOAD(dst + (op + done + i - offset));
    done += piece;
  }
}

// LZ4's length bytes (each adds its value; the first one below 255 is the last), 64 at a time.  false: the stream ends
// inside them (ip then lies at its end, as a byte-by-byte reader leaves it).
__device__ __forceinline__ bool lw_length_bytes(const PQD_GLOBAL uint8_t* src, uint32_t& ip, uint32_t in_end, uint64_t& sum, uint32_t lane) {
  uint32_t dog6 = 0;
  for (;;) {
    LW_DOG(dog6, 6)
    const bool inside = ip + lane < in_end;
    const uint32_t b = inside ? (uint32_t)src[ip + lane] : 0u;
    const uint64_t ends = __builtin_amdgcn_ballot_w64(!inside || b != 255u);
    if (!ends) {
      sum += 64u * 255u;
      ip += 64u;
      continue;
    }
    const uint32_t at = lw_uni((uint32_t)__builtin_ctzll(ends));
    sum += 255u * at;
    if (ip + at >= in_end) {
      ip = in_end;
      return false;
    }
    sum += (uint32_t)__builtin_amdgcn_readlane((int)b, (int)at);
    ip += at + 1u;
    return true;
  }
}

// false: the sequence breaks a rule (nothing of it is written; ip / op stay in front of it... LZ4: behind the literals
// already moved, as the format's decoders leave them)
template <int CODEC>
__device__ __forceinline__ bool lw_one(const PQD_GLOBAL uint8_t* src, PQD_GLOBAL uint8_t* dst, uint32_t& ip, uint32_t& op, uint32_t in_end,
                                       uint32_t cap, uint32_t lane) {
  if (CODEC == PQD_CODEC_SNAPPY) {
    const uint32_t tag = lw_byte(src, ip), left = in_end - ip, kind = tag & 3u;
    if (kind == 0u) {
      uint64_t l = tag >> 2;
      uint32_t hdr = 1;
      if (l >= 60u) {
        const uint32_t nb = (uint32_t)l - 59u;
        if (left < 1u + nb) return false;
        l = 0;
        for (uint32_t k = 0; k < nb; ++k) l |= (uint64_t)lw_byte(src, ip + 1u + k) << (8u * k);
        hdr = 1u + nb;
      }
      const uint64_t len = l + 1u;
      if (len > left - hdr || len > cap - op) return false;
      lw_copy_literal(src + ip + hdr, dst + op, (uint32_t)len, lane);
      op += (uint32_t)len;
      ip += hdr + (uint32_t)len;
    } else {
      const uint32_t hdr = kind == 1u ? 2u : kind == 2u ? 3u : 5u;
      if (left < hdr) return false;
      uint32_t rest = 0;
      for (uint32_t k = 1; k < hdr; ++k) rest |= lw_byte(src, ip + k) << (8u * (k - 1u));
      const uint32_t len = kind == 1u ? 4u + ((tag >> 2) & 7u) : 1u + (tag >> 2);
      const uint32_t offset = kind == 1u ? ((tag >> 5) << 8) | (rest & 0xFFu) : rest;
      if (offset == 0u || offset > op || len > cap - op) return false;
      lw_copy_match(dst, op, offset, len, lane);
      op += len;
      ip += hdr;
    }
    return true;
  }
  const uint32_t token = lw_byte(src, ip);
  ++ip;
  uint64_t lit = token >> 4;
  if (lit == 15u && !lw_length_bytes(src, ip, in_end, lit, lane)) return false;
  if (lit > in_end - ip || lit > cap - op) return false;
  lw_copy_literal(src + ip, dst + op, (uint32_t)lit, lane);
  op += (uint32_t)lit;
  ip += (uint32_t)lit;
  if (ip == in_end) return true;  // the last sequence has no match
  if (in_end - ip < 2u) return false;
  const uint32_t offset = lw_byte(src, ip) | (lw_byte(src, ip + 1u) << 8);
  ip += 2u;
  uint64_t mlen = token & 15u;
  if (mlen == 15u && !lw_length_bytes(src, ip, in_end, mlen, lane)) return false;
  mlen += 4u;
  if (offset == 0u || offset > op || mlen > cap - op) return false;
  lw_copy_match(dst, op, offset, (uint32_t)mlen, lane);
  op += (uint32_t)mlen;
  return true;
}

// Snappy's preamble: the uncompressed length as a varint32.  false: it is broken (ip stays 0).
__device__ __forceinline__ bool lw_snappy_preamble(const PQD_GLOBAL uint8_t* src, uint32_t in_end, uint32_t& ip, uint64_t& ulen) {
  uint32_t shift = 0, at = 0;
  ulen = 0;
  while (at < in_end && shift <= 28u) {
    const uint32_t b = lw_byte(src, at);
    ++at;
    ulen |= (uint64_t)(b & 0x7Fu) << shift;
    if (!(b & 0x80u)) {
      if (ulen > 0xFFFFFFFFull) return false;
      ip = at;
      return true;
    }
    shift += 7u;
  }
  return false;
}

// (with four waves: registers for seven waves per SIMD, so that seven pages a CU are in flight)
template <int CODEC, uint32_t WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES == 4u ? 7 : 1) void k_lz_pages(const pqd_decompress_desc* __restrict__ descs,
                                                            pqd_page_result* __restrict__ results, uint32_t n_pages) {
  __shared__ LwLds L;
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_decompress_desc d = descs[page];
  if (d.codec != (uint32_t)CODEC) return;  // another kernel's page
  const uint32_t tid = threadIdx.x;
  if (d.src_len > 0x7fffffffu) {
    if (tid == 0) {
      pqd_page_result r = {PQD_BAD_ARG, 0u, 0u, 0u};
      results[page] = r;
    }
    return;
  }
  const PQD_GLOBAL uint8_t* src = (const PQD_GLOBAL uint8_t*)d.src;
  PQD_GLOBAL uint8_t* dst = (PQD_GLOBAL uint8_t*)d.dst;
  const uint32_t cap = d.dst_len, in_end = d.src_len;
  uint32_t ip = 0, op = 0;
  int32_t status = PQD_OK;
  uint64_t ulen = 0;
  if (CODEC == PQD_CODEC_SNAPPY && !lw_snappy_preamble(src, in_end, ip, ulen)) status = PQD_DECODE;
#ifdef LW_PROF
  uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t prof_t = __builtin_readcyclecounter();
#endif
  if (status == PQD_OK) {
    uint32_t dog7 = 0;
    while (ip < in_end) {
      LW_DOG(dog7, 7)
      if (!lw_window<CODEC, WAVES>(L, src, dst, ip, op, in_end, cap, tid LW_PROF_PASS)) continue;
      if (ip >= in_end) break;
      // one sequence by the first wave; the others learn where it ended
      if (tid < 64u) {
        const bool fine = lw_one<CODEC>(src, dst, ip, op, in_end, cap, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
          L.after_ip = ip;
          L.after_op = op;
          L.after_failed = fine ? 0u : 1u;
        }
      }
      __syncthreads();
      ip = L.after_ip;
      op = L.after_op;
      const bool failed = L.after_failed != 0u;
      __syncthreads();
      if (failed) {
        status = PQD_DECODE;
        break;
      }
    }
    // the stream produces what its preamble announced (Snappy), and the page what its header announced
    if (status == PQD_OK && ((CODEC == PQD_CODEC_SNAPPY && op != ulen) || op != cap)) status = PQD_DECODE;
  }
  if (tid == 0) {
    pqd_page_result r = {status, op, ip, 0u};
    results[page] = r;
  }
#ifdef LW_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0 && cap >= 64u)
    for (int k = 0; k < 8; ++k) ((PQD_GLOBAL uint64_t*)d.dst)[k] = prof[k];
#endif
}

// The same with the parse of a window and the execution of the window before it going on at once: the first wave
// parses (and stages its window by itself), the other WAVES - 1 execute; the two meet at a workgroup barrier once per
// window.  A sequence that has to be taken by itself waits until everything in front of it has been executed.
template <int CODEC, uint32_t WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_lz_pipe_pages(const pqd_decompress_desc* __restrict__ descs,
                                                                 pqd_page_result* __restrict__ results, uint32_t n_pages) {
  __shared__ LwPipeLds L;
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_decompress_desc d = descs[page];
  if (d.codec != (uint32_t)CODEC) return;  // another kernel's page
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (d.src_len > 0x7fffffffu) {
    if (tid == 0) {
      pqd_page_result r = {PQD_BAD_ARG, 0u, 0u, 0u};
      results[page] = r;
    }
    return;
  }
  const PQD_GLOBAL uint8_t* src = (const PQD_GLOBAL uint8_t*)d.src;
  PQD_GLOBAL uint8_t* dst = (PQD_GLOBAL uint8_t*)d.dst;
  const uint32_t cap = d.dst_len, in_end = d.src_len;
  uint32_t ip = 0, op = 0;
  int32_t status = PQD_OK;
  uint64_t ulen = 0;
  if (CODEC == PQD_CODEC_SNAPPY && !lw_snappy_preamble(src, in_end, ip, ulen)) status = PQD_DECODE;
#ifdef LW_PROF
  uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t prof_t = __builtin_readcyclecounter();
#endif
  if (tid == 0) {
    L.arrived = 0;
    L.abort = 0;
  }
  __syncthreads();
  LwExecWaves bar = {&L.arrived, &L.abort, 0u, WAVES - 1u, lane};
  bool parse_on = true;     // false: a window stopped in front of a sequence that is to be taken by itself
  int exec_slot = -1;       // the parsed window that has not been executed yet
  uint32_t parse_slot = 0;
  uint32_t dog7 = 0;
  while (status == PQD_OK) {
    LW_DOG(dog7, 7)
    const bool do_parse = parse_on && ip < in_end;
    if (!do_parse && exec_slot < 0) {
      if (parse_on || ip >= in_end) break;  // the stream is used up
      // one sequence by the first wave (everything in front of it is in memory); the others learn where it ended
      if (wave == 0) {
        const bool fine = lw_one<CODEC>(src, dst, ip, op, in_end, cap, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
          L.after_ip = ip;
          L.after_op = op;
          L.after_failed = fine ? 0u : 1u;
        }
      }
      __syncthreads();
      ip = L.after_ip;
      op = L.after_op;
      if (L.after_failed) status = PQD_DECODE;
      __syncthreads();
      parse_on = true;
      continue;
    }
    if (wave == 0) {
      if (do_parse) {
        lw_stage(L.win(parse_slot), src, ip, lw_staged_end(ip, in_end), lane * 16u, 1024u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        LW_STAMP(0)
        lw_parse_window<CODEC>(L, parse_slot, ip, op, in_end, cap, lane LW_PROF_PASS);
      }
    } else if (exec_slot >= 0) {
      lw_execute<WAVES - 1u>(L, (uint32_t)exec_slot, src, dst, wave - 1u, lane, bar, nullptr, 0u LW_PROF_PASS);
    }
    __syncthreads();
    exec_slot = -1;
    if (do_parse) {
      const LwWindow W = L.window(parse_slot);
      ip = W.new_ip;
      op = W.out_end;
      if (W.stopped) parse_on = false;
      exec_slot = (int)parse_slot;
      parse_slot ^= 1u;
    }
    if (L.abort) status = PQD_DECODE;
  }
  // the stream produces what its preamble announced (Snappy), and the page what its header announced
  if (status == PQD_OK && ((CODEC == PQD_CODEC_SNAPPY && op != ulen) || op != cap)) status = PQD_DECODE;
  if (tid == 0) {
    pqd_page_result r = {status, op, ip, 0u};
    results[page] = r;
  }
#ifdef LW_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (cap >= 64u) {
    if (tid == 0)
      for (int k = 0; k < 4; ++k) ((PQD_GLOBAL uint64_t*)d.dst)[k] = prof[k];
    if (tid == 64u)
      for (int k = 4; k < 8; ++k) ((PQD_GLOBAL uint64_t*)d.dst)[k] = prof[k];
  }
#endif
}

// The same with the three serial steps of the parse and the execution going on at once, each on the window behind
// the next step's: table (first wave; SPLIT: its upper half by the first, its lower half a step later by the second),
// chain, elements, execution (all other waves); a workgroup barrier once per window.  The chain may leave a window far
// behind the next one (a long literal): the tables made meanwhile are then of no use, and the next one is made where
// the chain arrived.  The chain runs ahead of the check of the sequences it passes (the listing of the elements): when
// that stops in front of a sequence, what was chained behind it does not count.
template <int CODEC, uint32_t WAVES, bool SPLIT>
__global__ __launch_bounds__(WAVES * 64) void k_lz_staged_pages(const pqd_decompress_desc* __restrict__ descs,
                                                                   pqd_page_result* __restrict__ results, uint32_t n_pages) {
  using Lds = LwStagedLdsT<SPLIT>;
  constexpr uint32_t PARSERS = SPLIT ? 4u : 3u, WS = Lds::WIN_SLOTS, PS = Lds::PLACE_SLOTS;
  constexpr uint32_t W_LOWER = 1u, W_CHAIN = PARSERS - 2u, W_ELEMS = PARSERS - 1u;  // (W_LOWER only with SPLIT)
  extern __shared__ __attribute__((aligned(16))) uint8_t lw_staged_lds[];
  Lds& L = *reinterpret_cast<Lds*>(lw_staged_lds);
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_decompress_desc d = descs[page];
  if (d.codec != (uint32_t)CODEC) return;  // another kernel's page
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (d.src_len > 0x7fffffffu) {
    if (tid == 0) {
      pqd_page_result r = {PQD_BAD_ARG, 0u, 0u, 0u};
      results[page] = r;
    }
    return;
  }
  const PQD_GLOBAL uint8_t* src = (const PQD_GLOBAL uint8_t*)d.src;
  PQD_GLOBAL uint8_t* dst = (PQD_GLOBAL uint8_t*)d.dst;
  const uint32_t cap = d.dst_len, in_end = d.src_len;
  uint32_t ip = 0, op = 0;  // behind the last window whose elements are listed
  int32_t status = PQD_OK;
  uint64_t ulen = 0;
  if (CODEC == PQD_CODEC_SNAPPY && !lw_snappy_preamble(src, in_end, ip, ulen)) status = PQD_DECODE;
#ifdef LW_PROF
  uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t prof_t = __builtin_readcyclecounter();
#endif
  if (tid == 0) {
    L.arrived = 0;
    L.abort = 0;
    L.n_ready[0] = L.n_ready[1] = 0;
  }
  __syncthreads();
  LwExecWaves bar = {&L.arrived, &L.abort, 0u, WAVES - PARSERS, lane};
  uint32_t cip = ip, cop = op;  // where the chain has arrived
  bool parse_on = true;     // false: a window stopped in front of a sequence that is to be taken by itself
  bool tab_valid = false;   // a table is made and not used yet: that of window tab_grid, the tab_seq-th begun
  uint32_t tab_grid = 0, tab_seq = 0;
  bool hi_valid = false;    // (SPLIT) a table's upper half is made
  uint32_t hi_grid = 0, hi_seq = 0;
  bool chain_valid = false;  // a window is chained and its elements are not listed yet
  uint32_t a_seq = 0, b_seq = 0, c_seq = 0;  // tables begun, windows chained, windows listed
  int x_slot = -1;          // the listed window that has not been executed yet
  uint32_t dog7 = 0;
  while (status == PQD_OK) {
    LW_DOG(dog7, 7)
    const bool more = parse_on && cip < in_end;
    const uint32_t need = cip / LW_WIN;
    // tables (whole ones, half ones) that do not lie where the chain goes on are of no use
    if (tab_valid && tab_grid != need) tab_valid = false;
    if (hi_valid && hi_grid != need + (tab_valid ? 1u : 0u)) hi_valid = false;
    const bool b_work = more && tab_valid;
    const bool lo_work = more && hi_valid;
    // the table to begin: of the window behind those that are on their way
    const uint32_t a_grid = need + (tab_valid ? 1u : 0u) + (hi_valid ? 1u : 0u);
    const bool a_work = more && a_grid * LW_WIN < in_end;
    const bool c_work = chain_valid;
    if (!a_work && !lo_work && !b_work && !c_work && x_slot < 0) {
      if (parse_on || ip >= in_end) break;  // the stream is used up
      // one sequence by the first wave (everything in front of it is in memory); the others learn where it ended
      if (wave == 0) {
        const bool fine = lw_one<CODEC>(src, dst, ip, op, in_end, cap, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
          L.after_ip = ip;
          L.after_op = op;
          L.after_failed = fine ? 0u : 1u;
        }
      }
      __syncthreads();
      ip = L.after_ip;
      op = L.after_op;
      if (L.after_failed) status = PQD_DECODE;
      __syncthreads();
      cip = ip;
      cop = op;
      parse_on = true;
      continue;
    }
    if (wave == 0) {
      if (a_work) {
        LW_TICK
        const uint32_t w0 = a_grid * LW_WIN;
        uint8_t* win = L.win(a_seq % WS);
        lw_stage(win, src, w0, lw_staged_end(w0, in_end), lane * 16u, 1024u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        LW_STAMP(0)
        lw_places<CODEC>(win, L.place(a_seq % PS), w0, in_end, lane, 31, SPLIT ? 17 : 1);
        LW_STAMP(1)
      }
    } else if (SPLIT && wave == W_LOWER) {
      if (lo_work) lw_places<CODEC>(L.win(hi_seq % WS), L.place(hi_seq % PS), hi_grid * LW_WIN, in_end, lane, 15, 1);
    } else if (wave == W_CHAIN) {
      if (b_work) {
        LW_TICK
        LwChainRecord& K = L.chain(b_seq & 1u);
        uint32_t next_ip, next_op;
        const LwChained c = lw_chain(L.place(tab_seq % PS), tab_grid * LW_WIN, cip, cop, in_end, cap, lane, next_ip, next_op);
        K.entry[lane] = c.entry;
        K.first_elem[lane] = c.first_elem;
        K.first_out[lane] = c.first_out;
        if (lane == 0) {
          K.w0 = tab_grid * LW_WIN;
          K.win_slot = tab_seq % WS;
          K.ip = cip;
          K.op = cop;
          K.next_ip = next_ip;
          K.next_op = next_op;
        }
        LW_STAMP(2)
      }
    } else if (wave == W_ELEMS) {
      if (c_work) {
        LW_TICK
        const LwChainRecord& K = L.chain((b_seq - 1u) & 1u);
        const LwChained c = {K.entry[lane], K.first_elem[lane], K.first_out[lane]};
        lw_elements<CODEC>(L.win(K.win_slot), c, L.elems(c_seq & 1u), &L.window(c_seq & 1u), K.win_slot, K.w0, K.ip, K.op, in_end, cap, lane);
        if (Lds::READY_TILES) {  // the masks of the window's first tiles, so that the executing waves need not make them
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const LwWindow W = L.window(c_seq & 1u);
          constexpr uint32_t TILE = lw_tile_bytes(WAVES - PARSERS);
          uint32_t e_lo = 0, n = 0;
          for (uint32_t tb = W.o0; tb < W.out_end && n < Lds::READY_TILES; tb += TILE, ++n)
            lw_tile_masks(L.elems(c_seq & 1u), W.n_elems, W.out_end, tb, TILE, e_lo, L.parse_bits, L.ready(c_seq & 1u)[n], lane);
          if (lane == 0) L.n_ready[c_seq & 1u] = n;
        }
        LW_STAMP(3)
      }
    } else if (x_slot >= 0) {
      LW_TICK
      lw_execute<WAVES - PARSERS>(L, (uint32_t)x_slot, src, dst, wave - PARSERS, lane, bar, L.ready((uint32_t)x_slot),
                                  L.n_ready[x_slot] LW_PROF_PASS);
    }
    __syncthreads();
    x_slot = -1;
    bool stopped = false, astray = false;
    if (c_work) {
      const LwWindow W = L.window(c_seq & 1u);
      ip = W.new_ip;
      op = W.out_end;
      stopped = W.stopped != 0u;
      // (at the stream's end the chain does not say where it left: the place behind the last sequence is none)
      const LwChainRecord& KC = L.chain((b_seq - 1u) & 1u);
      astray = W.new_ip != KC.next_ip || W.out_end != KC.next_op;
      x_slot = (int)(c_seq & 1u);
      ++c_seq;
      chain_valid = false;
    }
    if (b_work) {
      const LwChainRecord& K = L.chain(b_seq & 1u);
      cip = K.next_ip;
      cop = K.next_op;
      chain_valid = true;
      ++b_seq;
      tab_valid = false;
    }
    if (SPLIT) {
      if (lo_work) {  // (the table before it has just been chained, or was of no use)
        tab_valid = true;
        tab_grid = hi_grid;
        tab_seq = hi_seq;
        hi_valid = false;
      }
      if (a_work) {
        hi_valid = true;
        hi_grid = a_grid;
        hi_seq = a_seq;
        ++a_seq;
      }
    } else if (a_work) {
      tab_valid = true;
      tab_grid = a_grid;
      tab_seq = a_seq;
      ++a_seq;
    }
    if (stopped || astray) {  // what was chained and tabled behind the stop does not count
      if (stopped) parse_on = false;
      chain_valid = false;
      tab_valid = false;
      hi_valid = false;
      cip = ip;
      cop = op;
    }
    if (L.abort) status = PQD_DECODE;
  }
  // the stream produces what its preamble announced (Snappy), and the page what its header announced
  if (status == PQD_OK && ((CODEC == PQD_CODEC_SNAPPY && op != ulen) || op != cap)) status = PQD_DECODE;
  if (tid == 0) {
    pqd_page_result r = {status, op, ip, 0u};
    results[page] = r;
  }
#ifdef LW_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (cap >= 64u) {
    if (tid == 0)
      for (int k = 0; k < 2; ++k) ((PQD_GLOBAL uint64_t*)d.dst)[k] = prof[k];
    if (tid == W_CHAIN * 64u) ((PQD_GLOBAL uint64_t*)d.dst)[2] = prof[2];
    if (tid == W_ELEMS * 64u) ((PQD_GLOBAL uint64_t*)d.dst)[3] = prof[3];
    if (tid == PARSERS * 64u)
      for (int k = 4; k < 8; ++k) ((PQD_GLOBAL uint64_t*)d.dst)[k] = prof[k];
  }
#endif
}

// stored bytes (an uncompressed codec, the level bytes of a V2 page, a V2 page with is_compressed = false)
// A descriptor whose codec is neither that nor the codec of the call has no kernel in this launch sequence: it
// gets PQD_BAD_ARG here, so that its result is never what the memory held before.
__global__ __launch_bounds__(256) void k_stored_pages(const pqd_decompress_desc* __restrict__ descs,
                                                      pqd_page_result* __restrict__ results, uint32_t n_pages,
                                                      uint32_t call_codec) {
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_decompress_desc d = descs[page];
  if (d.codec != PQD_CODEC_UNCOMPRESSED) {
    if (d.codec != call_codec && threadIdx.x == 0) {
      pqd_page_result r = {PQD_BAD_ARG, 0u, 0u, 0u};
      results[page] = r;
    }
    return;
  }
  const uint32_t n = d.src_len < d.dst_len ? d.src_len : d.dst_len;
  const uint32_t pieces = n >> 4;
  for (uint32_t i = threadIdx.x; i < pieces; i += 256u) {
    u32x4 v;
    __builtin_memcpy(&v, d.src + (size_t)i * 16, 16);
    __builtin_memcpy(d.dst + (size_t)i * 16, &v, 16);
  }
  for (uint32_t i = (pieces << 4) + threadIdx.x; i < n; i += 256u) d.dst[i] = d.src[i];
  if (threadIdx.x == 0) {
    pqd_page_result r = {d.src_len == d.dst_len ? PQD_OK : PQD_DECODE, n, n, 0u};
    results[page] = r;
  }
}

// PIPE: 0 = the waves parse, then execute; 1 = parse and execution at once; 2 = table, chain, elements and execution at once
template <int CODEC, uint32_t WAVES, int PIPE>
static hipError_t launch_lz_kernel(const pqd_decompress_desc* descs, pqd_page_result* results, uint32_t n_pages, hipStream_t stream) {
  if (PIPE == 2) {
    // (more than 64 KB of LDS: asked for once per process and kernel)
    // (sixteen waves: a CU to themselves, the table's halves by two waves)
    constexpr bool SPLIT = WAVES == 16u;
    static const hipError_t lds_fine = hipFuncSetAttribute((const void*)k_lz_staged_pages<CODEC, WAVES, SPLIT>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LwStagedLdsT<SPLIT>));
    if (lds_fine != hipSuccess) return lds_fine;
    hipLaunchKernelGGL((k_lz_staged_pages<CODEC, WAVES, SPLIT>), dim3(n_pages), dim3(WAVES * 64), sizeof(LwStagedLdsT<SPLIT>), stream, descs,
                       results, n_pages);
  } else if (PIPE == 1) {
    hipLaunchKernelGGL((k_lz_pipe_pages<CODEC, WAVES>), dim3(n_pages), dim3(WAVES * 64), 0, stream, descs, results, n_pages);
  } else {
    hipLaunchKernelGGL((k_lz_pages<CODEC, WAVES>), dim3(n_pages), dim3(WAVES * 64), 0, stream, descs, results, n_pages);
  }
  return hipGetLastError();
}

template <uint32_t WAVES, int PIPE>
static hipError_t launch_lz_waves(const pqd_decompress_desc* descs, pqd_page_result* results, uint32_t n_pages, int codec, hipStream_t stream) {
  if (codec == PQD_CODEC_SNAPPY) return launch_lz_kernel<PQD_CODEC_SNAPPY, WAVES, PIPE>(descs, results, n_pages, stream);
  if (codec == PQD_CODEC_LZ4_RAW) return launch_lz_kernel<PQD_CODEC_LZ4_RAW, WAVES, PIPE>(descs, results, n_pages, stream);
  return hipSuccess;
}

static hipError_t launch_lz(const pqd_decompress_desc* descs, pqd_page_result* results, uint32_t n_pages, int codec, hipStream_t stream) {
  // PQD_LZ_WAVES = 4, 8 or 16 fixes the waves per page, PQD_LZ_PIPE = 0 / 1 / 2 the kind of kernel (testing)
  const char* env = getenv("PQD_LZ_WAVES");
  const int forced = env ? atoi(env) : 0;
  const char* penv = getenv("PQD_LZ_PIPE");
  const int pipe = penv ? atoi(penv) : n_pages <= LW_PIPE8_PAGES ? 2 : n_pages <= LW_PIPE4_PAGES ? 1 : 0;
  const uint32_t waves = forced == 4 || forced == 8 || forced == 16 ? (uint32_t)forced
                         : pipe == 0                               ? 4u
                         : n_pages <= LW_PIPE16_PAGES              ? 16u
                         : n_pages <= LW_PIPE8_PAGES               ? 8u
                                                                   : 4u;
  if (pipe == 2) {  // (eight waves at least: three of them parse)
    if (waves == 16u) return launch_lz_waves<16u, 2>(descs, results, n_pages, codec, stream);
    return launch_lz_waves<8u, 2>(descs, results, n_pages, codec, stream);
  }
  if (pipe == 1) {
    if (waves == 16u) return launch_lz_waves<16u, 1>(descs, results, n_pages, codec, stream);
    if (waves == 8u) return launch_lz_waves<8u, 1>(descs, results, n_pages, codec, stream);
    return launch_lz_waves<4u, 1>(descs, results, n_pages, codec, stream);
  }
  if (waves == 16u) return launch_lz_waves<16u, 0>(descs, results, n_pages, codec, stream);
  if (waves == 8u) return launch_lz_waves<8u, 0>(descs, results, n_pages, codec, stream);
  return launch_lz_waves<4u, 0>(descs, results, n_pages, codec, stream);
}

hipError_t launch_decompress_pages(const pqd_decompress_desc* descs, pqd_page_result* results, uint32_t n_pages, int codec,
                                   hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  if (codec == PQD_CODEC_ZSTD || codec == PQD_CODEC_GZIP) {
    hipError_t ez = codec == PQD_CODEC_ZSTD ? launch_zstd_pages(descs, results, n_pages, stream) : launch_gzip_pages(descs, results, n_pages, stream);
    if (ez != hipSuccess) return ez;
    hipLaunchKernelGGL(k_stored_pages, dim3(n_pages), dim3(256), 0, stream, descs, results, n_pages, (uint32_t)codec);
    return hipGetLastError();
  }
  hipError_t e = launch_lz(descs, results, n_pages, codec, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_stored_pages, dim3(n_pages), dim3(256), 0, stream, descs, results, n_pages, (uint32_t)codec);
  return hipGetLastError();
}

}  // namespace pqd
