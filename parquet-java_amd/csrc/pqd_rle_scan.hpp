// The serial part of the RLE / bit-packing hybrid page decoders that keep a window of the stream in LDS
// (k_rle.hip: the decoder that knows every rule of the reference; k_binary_fused.hip): one lane walks the run
// headers of the window and leaves a table of segments {first value, source, kind}; the other lanes then
// expand the window's values in parallel.  Follows RunLengthBitPackingHybridDecoder.readNext
// (parquet-column/.../values/rle/RunLengthBitPackingHybridDecoder.java:80-109).
#pragma once

#include "pqd_device.hpp"

namespace pqd {

constexpr uint32_t SEG_RLE = 0u, SEG_PACKED = 1u;

struct RleState {
  uint32_t pos;             // next unread byte of the stream (offset from desc.data)
  uint32_t produced;        // values expanded before the current window
  uint32_t window_values;   // values covered by the current segment table
  uint32_t nseg;
  uint32_t pending_groups;  // groups left of a bit-packed run whose header is already consumed
  int32_t scan_status;      // first failure met by the header walk, PQD_OK if none
  uint32_t scan_err_at;     // output index at which that failure happens
  uint32_t oob_at;          // smallest output index with id >= dict_len (atomicMin), ~0u if none
  uint32_t done;
  int32_t window_base;      // stream offset of LDS byte 0 (may be slightly negative)
};

// Header walk over the staged window; run by one lane.  Follows readNext() (:80-109).
template <int WINDOW_BYTES, int MAX_SEGS>
__device__ void rle_scan_window(RleState& st, const uint32_t* win, const uint8_t* data, uint32_t* seg_start,
                                uint32_t* seg_src, uint32_t* seg_kind, uint32_t w, uint32_t end, uint32_t n_values,
                                uint32_t nseg0 = 0, uint32_t vals0 = 0) {  // nseg0 segments of vals0 values are in the table already
  uint32_t pos = st.pos;
  const int32_t base = st.window_base;
  const uint32_t wend = (uint32_t)(base + WINDOW_BYTES);  // stream offset one past the window
  const uint32_t remaining = n_values - st.produced;
  uint32_t vals = vals0, nseg = nseg0, pending = st.pending_groups;
  int32_t status = PQD_OK;
  bool stop = false;
  while (!stop && nseg < MAX_SEGS && vals < remaining) {
    if (pending == 0) {
      if (pos >= end) {  // :81 "Reading past RLE/BitPacking stream."
        status = PQD_PAST_END;
        break;
      }
      if (nseg > 0 && pos + 16 > wend && wend < end) break;  // header may straddle the window: restage
      // BytesUtils.readUnsignedVarInt (parquet-common/.../BytesUtils.java:202-211)
      uint32_t header = 0, shift = 0, p = pos, nb = 0;
      for (;;) {
        if (p >= end) {
          status = PQD_EOF;
          break;
        }
        // A header of more than five bytes is no writer's work, but the reference reads on until a
        // byte without the continuation bit, its shifts wrapping at 32: such a header may leave the
        // staged window, so from its sixth byte on it is read from memory.
        uint32_t b = ++nb > 5 ? (uint32_t)data[p] : lds_byte(win, p - base);
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
        // RLE run: value on ceil(w/8) little-endian bytes, not masked to w (BytesUtils.java:124-142)
        uint32_t count = header >> 1;
        uint32_t nbytes = (w + 7) >> 3;
        if (end - p < nbytes) {
          status = PQD_EOF;
          break;
        }
        uint32_t value = 0;
        for (uint32_t i = 0; i < nbytes; ++i) value |= (nb > 5 ? (uint32_t)data[p + i] : lds_byte(win, p - base + i)) << (8 * i);
        p += nbytes;
        pos = p;
        // a zero-count run leaves currentCount negative: the value repeats for the rest of the page
        uint32_t take = (count == 0 || count > remaining - vals) ? remaining - vals : count;
        seg_start[nseg] = vals;
        seg_src[nseg] = value;
        seg_kind[nseg] = SEG_RLE;
        ++nseg;
        vals += take;
        continue;
      }
      uint32_t groups = header >> 1;
      if (groups == 0 || groups > 0x0FFFFFFFu) {  // empty buffer indexed / negative array size
        status = PQD_DECODE;
        break;
      }
      pending = groups;
      pos = p;
    }
    // bit-packed groups (8 values, w bytes each)
    const uint32_t want_groups = (remaining - vals + 7) >> 3;
    if (w == 0) {
      uint32_t g = pending < want_groups ? pending : want_groups;
      uint32_t take = g * 8 > remaining - vals ? remaining - vals : g * 8;
      seg_start[nseg] = vals;
      seg_src[nseg] = 0;
      seg_kind[nseg] = SEG_RLE;
      ++nseg;
      vals += take;
      pending = 0;
      continue;
    }
    const uint32_t avail = end > pos ? end - pos : 0u;      // real bytes left in the stream
    uint32_t g = pending;
    if (g > want_groups) g = want_groups;
    // with_data: groups holding at least one real byte; in_window: groups that lie wholly in the staged window.
    // Both are divisions by w (some forty instructions each for the one lane that runs this): they are only
    // worked out when g groups do not fit, which happens once per window at most.
    const uint32_t room = wend > pos ? wend - pos : 0u;
    const uint64_t g_bytes = (uint64_t)g * w;
    const bool fits = g != 0 && g_bytes <= avail && g_bytes <= room;
    const uint32_t with_data = fits ? g : (avail + w - 1) / w;
    const uint32_t in_window = fits ? g : room / w;
    if (with_data == 0) {
      // the stream ended inside the run: the rest of its buffer stays zero (:95-99)
      uint32_t take = g * 8 > remaining - vals ? remaining - vals : g * 8;
      seg_start[nseg] = vals;
      seg_src[nseg] = 0;
      seg_kind[nseg] = SEG_RLE;
      ++nseg;
      vals += take;
      pending = 0;
      continue;
    }
    if (g > with_data) g = with_data;
    if (g > in_window) g = in_window;
    if (g == 0) {
      stop = true;  // window exhausted: restage from pos
      break;
    }
    uint32_t take = g * 8 > remaining - vals ? remaining - vals : g * 8;
    seg_start[nseg] = vals;
    seg_src[nseg] = (uint32_t)(pos - base);  // LDS byte offset of the first group
    seg_kind[nseg] = SEG_PACKED;
    ++nseg;
    vals += take;
    uint32_t bytes = g * w;
    pos += bytes > avail ? avail : bytes;
    pending -= g;
  }
  seg_start[nseg] = vals;
  st.nseg = nseg;
  st.window_values = vals;
  st.pending_groups = pending;
  if (status != PQD_OK) {
    st.scan_status = status;
    st.scan_err_at = st.produced + vals;
    st.done = 1;
  } else if (st.produced + vals >= n_values) {
    // readNext consumed the whole payload of the last run when its header was read
    if (pending > 0 && w > 0) {
      uint64_t rest = (uint64_t)pending * w;
      uint32_t avail = end > pos ? end - pos : 0u;
      pos += rest > avail ? avail : (uint32_t)rest;
    }
    st.done = 1;
  }
  st.pos = pos;
}

}  // namespace pqd
