// Host-callable launchers of the page-decode kernels (one per kernel source file).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pqd.h"
#include "pqd_rle_plan.hpp"

namespace pqd {

hipError_t launch_unpack_run_i32(const uint8_t* in, uint64_t in_bytes, int32_t* out, uint64_t n_values,
                                 uint32_t w, hipStream_t stream);
hipError_t launch_unpack_run_i64(const uint8_t* in, uint64_t in_bytes, int64_t* out, uint64_t n_values,
                                 uint32_t w, hipStream_t stream);
// elem: 4 or 8; has_dict: PQD_ENC_RLE_DICTIONARY pages
// only_fallback_plans: nullptr = decode every page; else decode only pages whose plan is flagged for fallback
// (the plan kernel marks those pages in plans[page].flags and sets the flag *fallback_count)
// (pool / pool_chunks / cursor: planned dictionary pages only: the same launch expands the short chunks that the
// plan kernel took from the pool's end, DS_DIRECT_MAX in pqd_rle_plan.hpp, looking their ids up in memory)
hipError_t launch_rle_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages, int elem,
                            bool has_dict, const void* only_fallback_plans, const uint32_t* fallback_count,
                            hipStream_t stream, const void* pool = nullptr, uint32_t pool_chunks = 0,
                            const uint32_t* cursor = nullptr, uint32_t* zero = nullptr, uint32_t zero_words = 0);
// (zero / zero_words: words the launch sets to zero besides its work: the counter block of the call after this one)
// header walk of hybrid pages (plain ones, or dictionary pages with want_dict) -> chunks of run
// segments in `pool` (pqd_rle_plan.hpp): `plans` holds n_pages RlePagePlan records, `pool` pool_chunks
// chunks, `cursor` two zeroed counters (chunks taken from the pool's shared tail; pages left to the
// generic kernel)
// (dictionary pages of 8-byte values: `planes` is where dictionaries that need plane copies get their slot)
hipError_t launch_rle_plan_chunks(const pqd_page_desc* descs, pqd_page_result* results, void* plans, void* pool,
                                  uint32_t pool_chunks, uint32_t* cursor, uint32_t n_pages, bool want_dict, int elem,
                                  const DictPlanes& planes, bool try_small, hipStream_t stream, bool long_ok = false);
// Long pages of many short runs that the plan kernel marked PLAN_F_LONG (launch_rle_plan_chunks with long_ok calls it):
// a workgroup of 1024 threads per page; a page it gives up is marked for the left-over launch.
constexpr uint32_t WP_LONG_MAX_PAGES = 1024;  // calls of at most that many pages get the launch
hipError_t launch_whole_long_pages(const pqd_page_desc* descs, pqd_page_result* results, void* plans, uint32_t* cursor, uint32_t n_pages,
                                   bool has_dict, int elem, hipStream_t stream);
// small plain hybrid pages (stream <= 8 KiB, <= 32768 values, widths 1..24) decoded whole; sets plans[page].flags
// for every page (PLAN_F_SMALL_DONE where it decoded the page)
hipError_t launch_rle_small_pages(const pqd_page_desc* descs, pqd_page_result* results, void* plans, uint32_t n_pages,
                                  hipStream_t stream);
// plain hybrid pages: expansion of all chunks
hipError_t launch_rle_expand(const void* pool, uint32_t pool_chunks, const uint32_t* cursor, uint32_t n_pages,
                             hipStream_t stream);
// planned dictionary pages: expansion + sliced LDS gather
// (ev_start / ev_stop, if given: events that the launch itself records at the kernel's start and end)
hipError_t launch_dict_pages_sliced(pqd_page_result* results, const void* pool, uint32_t pool_chunks, const uint32_t* cursor,
                                    uint32_t n_pages, int elem, const uint8_t* plane_area, hipStream_t stream,
                                    hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
hipError_t launch_delta_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages, int elem,
                              hipStream_t stream);
// DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY: the length blocks into the page's work area (two result
// records in DELTA_BYTES_WORK_HEADER bytes, then the int32 lengths), then offsets + bytes
constexpr uint32_t DELTA_BYTES_WORK_HEADER = 64;
hipError_t launch_delta_length_blocks(const pqd_delta_bytes_desc* descs, uint32_t n_pages, hipStream_t stream);
hipError_t launch_delta_bytes(const pqd_delta_bytes_desc* descs, pqd_page_result* results, uint32_t n_pages, hipStream_t stream);
// PLAIN BYTE_ARRAY pages among the same descriptors (encoding PQD_ENC_PLAIN_BYTE_ARRAY; the other pages are left alone)
hipError_t launch_plain_byte_array(const pqd_delta_bytes_desc* descs, pqd_page_result* results, uint32_t n_pages, hipStream_t stream);
// SNAPPY / LZ4_RAW / ZSTD pages (one wave each) and stored parts (desc.codec PQD_CODEC_UNCOMPRESSED)
hipError_t launch_gzip_pages(const pqd_decompress_desc* descs, pqd_page_result* results, uint32_t n_pages, hipStream_t stream);
hipError_t launch_zstd_pages(const pqd_decompress_desc* descs, pqd_page_result* results, uint32_t n_pages, hipStream_t stream);
hipError_t launch_decompress_pages(const pqd_decompress_desc* descs, pqd_page_result* results, uint32_t n_pages, int codec,
                                   hipStream_t stream);
hipError_t launch_bss_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages,
                            hipStream_t stream);
hipError_t launch_bitpacked_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages,
                                  hipStream_t stream);
hipError_t launch_plain_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages, hipStream_t stream);
hipError_t launch_plain_boolean_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages, int elem,
                                      hipStream_t stream);
// decoded levels -> validity bitmap / value positions, null scatter, list offsets (one workgroup per page)
hipError_t launch_def_validity(const pqd_validity_desc* descs, pqd_page_result* results, uint32_t n_pages, hipStream_t stream);
hipError_t launch_scatter_nulls(const pqd_scatter_desc* descs, pqd_page_result* results, uint32_t n_pages, int elem,
                                hipStream_t stream);
hipError_t launch_rep_offsets(const pqd_offsets_desc* descs, pqd_page_result* results, uint32_t n_pages, hipStream_t stream);
// BINARY values gathered from their views into offsets + bytes, one workgroup per page
hipError_t launch_binary_fused(const pqd_page_desc* pages, const pqd_binary_page_desc* bins, const void* slots, uint32_t n_slots,
                               pqd_page_result* page_results, pqd_page_result* bin_results, uint32_t n_pages, hipStream_t stream,
                               hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
hipError_t launch_binary_pages(const pqd_binary_page_desc* descs, pqd_page_result* results, uint32_t n_pages,
                               hipStream_t stream);

}  // namespace pqd
