// Run plan of an RLE / bit-packing hybrid page: the output of the header-walk kernel
// (k_rle_plan_chunks) and the input of the expansion kernels (k_rle_expand, k_dict_pages_sliced).
//
// The hybrid grammar is sequential (each run header sits right behind the previous run's
// payload), so finding the runs is a pointer chase.  It is done once per page by a single wave,
// thousands of pages at a time, and leaves a table of segments in device scratch memory; the
// kernels that move the bulk of the bytes then start from the table and contain no serial code.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pqd {

// One segment = a run, or a piece of a bit-packed run, whose values are consecutive in the output.
struct RleSeg {
  uint32_t start4;  // (output index of the first value << 2) | SEG_PACKED_BIT | SEG_PARTIAL_BIT.
                    // With the flags below the index, start4 values compare like the indices.
  uint32_t src;     // bit-packed: stream offset of the first group; RLE: the repeated value
};

constexpr uint32_t SEG_PACKED_BIT = 1u;   // bit-packed
constexpr uint32_t SEG_PARTIAL_BIT = 2u;  // bit-packed and cut short by the end of the stream (missing bytes are zero)
constexpr uint32_t SEG_START_MAX = 0x3FFFFFFFu;

// The plan of a page is a list of CHUNKS taken from a pool in device memory.  A chunk is SEG_CHUNK
// 8-byte slots (2 KiB) and describes the output values [v0, v1) of one page: v0 and v1 are
// multiples of SEG_BLOCK (v1 may also be the page's end), and
//   slots 0 .. nseg-1          the segments; the first one may begin before v0
//   slot  nseg                 a sentinel whose start is v1 (so "start of the next segment" always exists)
//   slots SEG_HEAD_SLOT ..     the chunk's header (RleChunkHead): the consumers read nothing else of the page
//   slots SEG_TABLE_SLOT ..    one byte per block of SEG_BLOCK output values: the segment that holds the
//                              block's first value, block b starting at value v0 + b * SEG_BLOCK
//                              (a wave's unit row is one such block)
// Pool chunk p is reserved for page p (an empty chunk, v0 == v1, if the page yields nothing for the
// consumer); a page's further chunks come from the shared tail of the pool through one counter.
constexpr int SEG_CHUNK = 256;
constexpr int SEG_TABLE_SLOT = 240;
constexpr int SEG_BLOCK = 256;
constexpr int SEG_TABLE_BLOCKS = (SEG_CHUNK - SEG_TABLE_SLOT) * 8;  // 128 blocks = 32768 values
constexpr int PLAN_SEG_MAX_BYTES = 6144;  // payload bytes of one bit-packed segment
constexpr int SEG_HEAD_SLOT = 232;               // slots 232..239: RleChunkHead
constexpr int SEG_GEN_MAX = SEG_HEAD_SLOT - 1;   // segments per chunk, at most
constexpr uint32_t SEG_CHUNK_SPAN = SEG_TABLE_BLOCKS * SEG_BLOCK;  // values per chunk, at most

struct RleChunkHead {
  uint64_t data;        // the page's stream
  uint64_t out;         // the page's output (element 0)
  uint32_t v0, v1;      // output values covered
  uint32_t nseg;        // bits 0..15: segments; bits 16..31: 1 + the dictionary's plane slot (0: none), see DictPlanes
  uint32_t w;
  uint32_t data_len;
  uint32_t stream_end;  // bytes at or past it read as zero
  uint32_t page;
  uint32_t dict_len;    // dictionary pages: entries of the dictionary ...
  uint64_t dict;        // ... and where it is
  uint32_t reserved[2];  // [0] != 0: every segment is a whole bit-packed run of [0] values and [1] stream
                         // bytes, one right behind the other (k_rle_expand then needs no search)
};
static_assert(sizeof(RleChunkHead) == 64, "chunk header is 64 bytes");

// Dictionary pages: a chunk of at most DS_DIRECT_MAX values over a dictionary that would have to be swept through
// LDS is not gathered by the sliced kernel (a sweep costs the same whatever the chunk holds, about 10 us of a CU) but
// by k_dict_expand, which reads the dictionary from memory (a cache line per value at most): run-heavy pages, which
// are cut into several chunks of a few thousand values (a chunk holds 231 segments), and small pages.  Such chunks
// are taken from the END of the pool (the last one first), so the sliced kernel never sees them; if it is a page's
// first chunk, the page's own slot holds an empty chunk.
constexpr uint32_t DS_DIRECT_MAX = 8192;

// The pool of a call that plans dictionary pages: n_pages page slots, then the shared tail, of which the front half
// serves the chunks of the sliced gather (counter 0, growing upwards from slot n_pages) and the back half the chunks
// of k_dict_expand (counter 4, growing downwards from the last slot).  Plain hybrid pages use the whole tail from the
// front.  plan_front_limit: the first slot the front may NOT use (what the consumers of the front take as the pool's size).
__host__ __device__ inline uint32_t plan_front_limit(uint32_t pool_chunks, uint32_t n_pages, bool want_dict) {
  if (pool_chunks <= n_pages || !want_dict) return pool_chunks;
  return pool_chunks - (pool_chunks - n_pages) / 2u;
}

// The call's counter block (32-bit counters, zeroed per call):
//   0  chunks asked for at the front of the pool's shared tail (behind the n_pages page slots; granted below plan_front_limit)
//   1  non-zero: some page is left to the generic kernel (PLAN_F_FALLBACK in its plan record)
//   2  plane slots claimed in this call (DictPlanes::list)
//   3  non-zero: dictionary chunks that are neither regular nor listed exist (0: the batch takes the regular path
//      of k_dict_pages_sliced)
//   4  chunks asked for at the end of the pool (k_dict_expand's; granted down to plan_front_limit)
//   5  non-zero: some dictionary page is marked PLAN_F_WHOLE (the left-over launch looks for them)
//   6  irregular chunks of large dictionaries that were listed (while counter 3 is zero they are all listed: the
//      regular path skips them, the left-over launch expands them)
//   7 / 8  non-zero: some regular chunk has a dictionary that is swept through LDS / that stays resident there (the
//      regular path takes batches of one kind)
//   64 .. 64 + PLAN_IRREGULAR_LIST - 1   their pool slots
constexpr int PLAN_CTR_TAIL = 0, PLAN_CTR_FALLBACK = 1, PLAN_CTR_PLANES = 2, PLAN_CTR_IRREGULAR = 3, PLAN_CTR_DIRECT = 4, PLAN_CTR_WHOLE = 5,
              PLAN_CTR_IRREGULAR_LISTED = 6, PLAN_CTR_SWEPT = 7, PLAN_CTR_RESIDENT = 8,
              PLAN_CTR_LIST0 = 64;
constexpr uint32_t PLAN_IRREGULAR_LIST = 256;
constexpr uint32_t PLAN_COUNTER_BYTES = (PLAN_CTR_LIST0 + PLAN_IRREGULAR_LIST) * 4;

constexpr uint32_t PLAN_F_DONE = 1u;      // result already written by the plan kernel (empty / rejected page)
constexpr uint32_t PLAN_F_SMALL_DONE = 4u;  // decoded by k_rle_small_pages before the plan kernel ran (which turns it into PLAN_F_DONE)
constexpr uint32_t PLAN_F_WHOLE = 8u;     // a run-heavy dictionary page left unplanned: decoded whole by the left-over launch (pqd_whole_page.hpp)
constexpr uint32_t PLAN_F_LONG = 16u;     // a long page of many short runs, left unplanned: decoded by k_whole_long_pages (which marks what it gives up PLAN_F_FALLBACK)
constexpr uint32_t PLAN_F_FALLBACK = 2u;  // not for the planned kernels (too many segments, dictionary not sweepable): generic kernel

// Sliced dictionary gather: LDS slice size and the values one workgroup holds in registers per
// sweep.  The plan kernel uses them to decide whether sweeping the dictionary pays for a page.
constexpr uint32_t DS_SLICE_BYTES = 64 * 1024;
constexpr uint32_t DS_TILE_VALUES = 1024 * 20;
static_assert(DS_TILE_VALUES <= SEG_TABLE_BLOCKS * SEG_BLOCK, "the block table covers a whole tile");
constexpr uint32_t DS_MAX_PASSES = 32;

// Plane copies of 8-byte dictionaries that are swept through LDS (more than DS_SLICE_BYTES of entries,
// at most 65536 of them).  The sliced gather looks such a dictionary up four bytes at a time: slot s of
// the plane area holds the LOW dwords of all entries at byte 0 and the HIGH dwords at byte
// DP_PLANE_BYTES, so that a quarter of a plane (DP_QUARTER_ENTRIES entries, 64 KiB) is one contiguous
// piece for the LDS-DMA.  The plan kernel finds or claims the slot of a page's dictionary in a small
// open-addressing table keyed by the dictionary's address and length (`keys`, zeroed per call), and
// lists the slots it claimed (`list`, their number in counter 2 of the call's counter block);
// the plan kernel (dict_plane_piece) then fills them, and the chunk headers carry the slot to the gather.
constexpr uint32_t DP_PLANE_BYTES = 256 * 1024;
constexpr uint32_t DP_SLOT_BYTES = 2 * DP_PLANE_BYTES;
constexpr uint32_t DP_QUARTER_ENTRIES = 16384;
constexpr uint32_t DP_MAX_SLOTS = 512;
constexpr uint32_t DP_MAX_PROBES = 32;
struct DictPlanes {
  uint64_t* keys;    // n_slots entries: 0 = free, else dictionary address | (entries - 1) << 48
  uint32_t* list;    // slots claimed in this call
  uint8_t* area;     // n_slots x DP_SLOT_BYTES
  uint32_t n_slots;  // 0: no plane area (every 8-byte dictionary beyond one slice is gathered from memory)
};

// One piece (DP_PIECE_ENTRIES entries) of a claimed slot's plane copy, moved by `threads` lanes of a workgroup: item =
// (index into DictPlanes::list) * DP_PIECES + piece.  A lane moves four entries per step: two 16-byte loads, one
// 16-byte store per plane (whole 16-byte pieces: the planes are padded to them).
constexpr uint32_t DP_PIECE_ENTRIES = 2048;
constexpr uint32_t DP_PIECES = 65536 / DP_PIECE_ENTRIES;  // pieces of the largest dictionary
typedef uint32_t dp_u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void dict_plane_piece(const DictPlanes& planes, uint32_t item, uint32_t tid, uint32_t threads) {
  const uint32_t slot = planes.list[item / DP_PIECES];
  if (slot >= planes.n_slots) return;
  const uint64_t key = planes.keys[slot];
  const uint32_t entries = (uint32_t)(key >> 48) + 1u;
  const uint32_t e0 = (item % DP_PIECES) * DP_PIECE_ENTRIES;
  if (e0 >= entries) return;
  const uint32_t* src = reinterpret_cast<const uint32_t*>((uintptr_t)(key & 0xFFFFFFFFFFFFull));  // 16-byte aligned (plan kernel)
  uint32_t* lo = reinterpret_cast<uint32_t*>(planes.area + (size_t)slot * DP_SLOT_BYTES);
  uint32_t* hi = reinterpret_cast<uint32_t*>(planes.area + (size_t)slot * DP_SLOT_BYTES + DP_PLANE_BYTES);
  for (uint32_t e = e0 + 4u * tid; e < e0 + DP_PIECE_ENTRIES && e < entries; e += 4u * threads) {
    dp_u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u};
    if (e + 4u <= entries) {
      a = *reinterpret_cast<const dp_u32x4*>(src + 2 * (size_t)e);
      b = *reinterpret_cast<const dp_u32x4*>(src + 2 * (size_t)e + 4);
    } else {  // the dictionary's last one to three entries
      uint32_t t[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
      for (uint32_t j = 0; j < 2u * (entries - e); ++j) t[j] = src[2 * (size_t)e + j];
      a = dp_u32x4{t[0], t[1], t[2], t[3]};
      b = dp_u32x4{t[4], t[5], t[6], t[7]};
    }
    const dp_u32x4 l = {a.x, a.z, b.x, b.z}, h = {a.y, a.w, b.y, b.w};
    *reinterpret_cast<dp_u32x4*>(lo + e) = l;
    *reinterpret_cast<dp_u32x4*>(hi + e) = h;
  }
}

// Per-page record of the plan kernel.  `flags` belongs to the page with that index.
struct RlePagePlan {
  uint32_t flags;          // PLAN_F_*
  uint32_t reserved;
};
static_assert(sizeof(RlePagePlan) == 8, "plan record is 8 bytes");

}  // namespace pqd
