// Host-callable launchers of the page-decode kernels (one per kernel source file).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pqd.h"

namespace pqd {

hipError_t launch_unpack_run_i32(const uint8_t* in, uint64_t in_bytes, int32_t* out, uint64_t n_values,
                                 uint32_t w, hipStream_t stream);
hipError_t launch_unpack_run_i64(const uint8_t* in, uint64_t in_bytes, int64_t* out, uint64_t n_values,
                                 uint32_t w, hipStream_t stream);
// elem: 4 or 8; has_dict: PQD_ENC_RLE_DICTIONARY pages
hipError_t launch_rle_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages, int elem,
                            bool has_dict, hipStream_t stream);
hipError_t launch_delta_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages, int elem,
                              hipStream_t stream);
hipError_t launch_bss_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages,
                            hipStream_t stream);
hipError_t launch_bitpacked_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages,
                                  hipStream_t stream);

}  // namespace pqd
