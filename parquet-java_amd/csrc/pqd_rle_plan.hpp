// Run plan of an RLE / bit-packing hybrid page: the output of the header-walk kernel
// (k_rle_plan) and the input of the expansion kernels.
//
// The hybrid grammar is sequential (each run header sits right behind the previous run's
// payload), so finding the runs is a pointer chase.  It is done once per page by a single wave,
// thousands of pages at a time, and leaves a table of segments in device scratch memory; the
// kernels that move the bulk of the bytes then start from the table and contain no serial code.
#pragma once

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

// The pool holds one chunk of SEG_CHUNK 8-byte slots per page:
//   slots 0 .. nseg-1          the segments
//   slot  nseg                 a sentinel whose start is n_out (so "start of the next segment" always exists)
//   slots SEG_TABLE_SLOT ..    one byte per block of SEG_BLOCK output values: the segment that holds
//                              the block's first value (a wave's unit row is one such block)
constexpr int SEG_CHUNK = 256;
constexpr int SEG_TABLE_SLOT = 240;
constexpr int SEG_MAX = SEG_TABLE_SLOT - 1;  // segments a planned page may have; beyond: generic kernel
constexpr int SEG_BLOCK = 256;
constexpr int SEG_TABLE_BLOCKS = (SEG_CHUNK - SEG_TABLE_SLOT) * 8;  // 128 blocks = 32768 values
constexpr int PLAN_SEG_MAX_BYTES = 6144;  // payload bytes of one bit-packed segment

// Plain hybrid pages (levels, ids, booleans) of any length are planned as a list of such chunks,
// each covering the output values [v0, v1) of one page: v0 and v1 are multiples of SEG_BLOCK (v1
// may also be the page's end), the first segment may begin before v0, the sentinel's start is v1,
// table block b starts at value v0 + b * SEG_BLOCK, and the chunk carries its own header:
constexpr int SEG_HEAD_SLOT = 232;               // slots 232..239: RleChunkHead
constexpr int SEG_GEN_MAX = SEG_HEAD_SLOT - 1;   // segments per chunk of the general path
constexpr uint32_t SEG_CHUNK_SPAN = SEG_TABLE_BLOCKS * SEG_BLOCK;  // values per chunk, at most

struct RleChunkHead {
  uint64_t data;        // the page's stream
  uint64_t out;         // the page's output (element 0)
  uint32_t v0, v1;      // output values covered
  uint32_t nseg;
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

constexpr uint32_t PLAN_F_DONE = 1u;      // result already written by the plan kernel (empty / rejected page)
constexpr uint32_t PLAN_F_FALLBACK = 2u;  // not for the planned kernels (too many segments, dictionary not sweepable): generic kernel

// Sliced dictionary gather: LDS slice size and the values one workgroup holds in registers per
// sweep.  The plan kernel uses them to decide whether sweeping the dictionary pays for a page.
constexpr uint32_t DS_SLICE_BYTES = 64 * 1024;
constexpr uint32_t DS_TILE_VALUES = 1024 * 20;
static_assert(DS_TILE_VALUES <= SEG_TABLE_BLOCKS * SEG_BLOCK, "the block table covers a whole tile");
constexpr uint32_t DS_MAX_PASSES = 32;

// Everything the expansion kernel needs to know about a page (it does not read the descriptor).
// The segment chunk of page p is pool[p * SEG_CHUNK ...].
struct RlePagePlan {
  int32_t status;       // status of the header walk (PQD_OK, or the failure met at output index n_out)
  uint32_t n_out;       // values covered by the segments
  uint32_t consumed;    // stream bytes consumed (bytes_consumed of the result)
  uint32_t nseg;
  uint32_t w;           // bit width of the ids
  uint32_t flags;       // PLAN_F_*
  uint32_t stream_end;  // offset from `data` where the hybrid stream ends
  uint32_t data_len;
  uint64_t data;        // copies of the descriptor's pointers
  uint64_t out;
  uint64_t dict;
  uint32_t dict_len;
  uint32_t reserved;
};
static_assert(sizeof(RlePagePlan) == 64, "plan record is 64 bytes");

}  // namespace pqd
