// Page decompression on the device: raw Snappy and the LZ4 block format (and stored bytes: a copy).
//
// Replaces the step before initFromPage in ColumnChunkPageReadStore.readPage (parquet-hadoop/.../hadoop/
// ColumnChunkPageReadStore.java:151-181 V1, :232-262 V2: decompressor.decompress(bytes, uncompressed size)),
// i.e. SnappyDecompressor.uncompress -> org.xerial.snappy.Snappy.uncompress (codec/SnappyDecompressor.java:
// 27-35) and Lz4RawDecompressor.uncompress -> io.airlift.compress.lz4.Lz4Decompressor (codec/
// Lz4RawDecompressor.java:29-49).  Both formats are a byte-serial stream of "copy these literal bytes" and
// "repeat `length` bytes from `offset` back" elements, so a page is one serial chain; pages are
// independent.  One wave per page.  Snappy is taken 64 input bytes at a time (snappy_window below: every byte
// position is tried as an element header, the real chain is found by lane exchanges, literal bytes move in
// one step, copies one after the other); LZ4 sequences, and Snappy streams in which something is out of
// order, go element by element: the headers are decoded wave-uniformly (scalar registers, read from a
// 16-byte window of the input kept there), the 64 lanes move the bytes.  The input is staged in LDS in 2 KiB
// chunks, the next chunk already on its way in registers; the last 16 KiB of output are kept in an LDS ring,
// which is where most matches read from; a match from further back reads the bytes back from L2.
// Lane l of a match step takes the byte `offset` back, or for offsets below 64 the byte (l mod offset) -
// offset back of the step's start, which is the same byte of the repeating pattern and always already
// written.
// Not HBM-bound: a page is a serial chain and a lone wave issues an instruction every ~17 cycles here
// (branches, instruction fetch), about 20 MB/s per page; throughput is the number of pages in flight, which
// LDS sets (20.5 KiB per page: seven per CU, 1792 per chip; 30 GB/s of text at that many pages).  Measured
// and dropped: buffering the output in the ring and writing it in 16-byte pieces instead of storing each
// element's bytes (slower: the stores are not what the chain waits for); a 32 KiB ring (fewer pages in
// flight, and reading far matches back from L2 costs little once no system-wide fence is involved).
#include <cstdlib>

#include "pqd_device.hpp"
#include "pqd_kernels.hpp"
#include "pqd_lz_common.hpp"

namespace pqd {

// Snappy, 64 input bytes at a time.  Every lane reads the byte at ip + lane as if an element started there
// (kind, header bytes, lengths, offset) and notes where the element after it would start.  Six rounds of
// "and from there?" between the lanes give the 2^k-th successor of every position, from which lane i works
// out the i-th element of the real chain that starts at ip: the elements of the window, in order, one per
// lane.  A prefix sum of their output lengths places them; literal bytes are then moved by all lanes at
// once (a lane per input byte), copies one after the other (they may read what the one before wrote), each
// by the whole wave.  Returns false, with nothing done, when an element of the window breaks a rule
// (offset 0 or beyond the output, bytes beyond the stream or the page): the caller then takes the
// one-element-at-a-time path, which reports the failure the way the format's decoder does.
constexpr uint32_t SW_BAD = 0xFFFFFFFFu, SW_END = 0xFFFFFFFEu;

__device__ __forceinline__ bool snappy_window(InStage& in, OutRing& out, uint8_t* marks, uint32_t& ip, uint32_t& op,
                                              uint32_t in_end, uint32_t cap, uint32_t lane) {
  const uint32_t p = ip + lane;
  const uint32_t* st32 = reinterpret_cast<const uint32_t*>(in.lds);
  const uint32_t w0 = (p & (LZ_STAGE - 1u)) >> 2, sh = (p & 3u) * 8u;
  const uint32_t d0 = st32[w0], d1 = st32[(w0 + 1u) & (LZ_STAGE / 4u - 1u)], d2 = st32[(w0 + 2u) & (LZ_STAGE / 4u - 1u)];
  const uint32_t lo = funnel_shr(d0, d1, sh), hi = funnel_shr(d1, d2, sh);  // bytes p .. p+3, p+4 .. p+7
  const uint32_t tag = lo & 0xFFu, kind = tag & 3u, n6 = tag >> 2;
  const uint32_t rest = (lo >> 8) | (hi << 24);  // bytes p+1 .. p+4
  // as a literal
  const uint32_t nb = n6 >= 60u ? n6 - 59u : 0u;
  const uint32_t lit_field = nb == 4u ? rest : rest & ((1u << (8u * nb)) - 1u);
  const uint64_t lit_len = (uint64_t)(nb ? lit_field : n6) + 1u;
  // as a copy
  const uint32_t copy_hdr = kind == 1u ? 2u : kind == 2u ? 3u : 5u;
  const uint32_t copy_len = kind == 1u ? 4u + (n6 & 7u) : 1u + n6;
  const uint32_t offset = kind == 1u ? ((tag >> 5) << 8) | (rest & 0xFFu) : kind == 2u ? rest & 0xFFFFu : rest;
  const bool is_lit = kind == 0u;
  const uint32_t hdr = is_lit ? 1u + nb : copy_hdr;
  const uint64_t adv64 = is_lit ? hdr + lit_len : hdr;
  const bool fine = (uint64_t)p + adv64 <= in_end && (is_lit || offset != 0u) && lit_len < (1u << 28);  // (28 bits hold a length below)
  const uint32_t adv = (uint32_t)adv64;
  const uint32_t out_len = is_lit ? (uint32_t)lit_len : copy_len;
  // successor, counted from the window's first byte (a value of 64 or more: outside the window)
  uint32_t succ[6];
  succ[0] = (p >= in_end || !fine) ? SW_BAD : p + adv == in_end ? SW_END : lane + adv;  // the stream's end is no position of the chain
#pragma unroll
  for (int k = 1; k < 6; ++k) {
    const uint32_t further = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((succ[k - 1] & 63u) << 2), (int)succ[k - 1]);
    succ[k] = succ[k - 1] < 64u ? further : succ[k - 1];
  }
  // lane i: the i-th element of the chain from position 0
  uint32_t pos = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const uint32_t hop = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((pos & 63u) << 2), (int)succ[k]);
    if ((lane >> k) & 1u) pos = pos < 64u ? hop : pos;
  }
  const uint64_t inside = __builtin_amdgcn_ballot_w64(pos < 64u);  // lanes 0 .. T-1 (positions grow with i)
  const uint32_t T = (uint32_t)__builtin_popcountll(inside);         // at most 32: an element is two bytes or more
  const uint32_t leave = (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)T);  // where the chain leaves the window
  if (leave == SW_BAD) return false;
  const bool mine = lane < T;
  // the elements' fields, gathered from the lanes that parsed them
  const uint32_t src_lane = (pos & 63u) << 2;
  const uint32_t e_meta = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src_lane, (int)(out_len | (hdr << 28) | (is_lit ? 0x80000000u : 0u)));
  const uint32_t e_off = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src_lane, (int)offset);
  const uint32_t e_out = mine ? e_meta & 0x0FFFFFFFu : 0u;  // (a literal of 2^28 bytes or more does not fit the stream: `fine` saw to it)
  const bool e_lit = (e_meta >> 31) != 0;
  const uint32_t inc = wave_scan_inclusive(e_out);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
  const uint32_t e_op = op + inc - e_out;  // where the element's output starts
  if ((uint64_t)op + total > cap) return false;
  if (__builtin_amdgcn_ballot_w64(mine && !e_lit && e_off > e_op)) return false;
  // ---- literal bytes, a lane per input byte: the element a byte belongs to is the last one that starts at
  // or before it
  __syncthreads();
  marks[lane] = 0;
  __syncthreads();
  if (mine) marks[pos] = 1;
  __syncthreads();
  const uint64_t starts = __builtin_amdgcn_ballot_w64(marks[lane] != 0);
  const uint32_t rank = (uint32_t)__builtin_popcountll(starts & ((2ull << lane) - 1ull)) - 1u;  // lane 0 starts an element
  const uint32_t o_pos = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(rank << 2), (int)pos);
  const uint32_t o_meta = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(rank << 2), (int)e_meta);
  const uint32_t o_op = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(rank << 2), (int)e_op);
  {
    const uint32_t first = o_pos + ((o_meta >> 28) & 7u);  // the literal's first byte
    if ((o_meta >> 31) && lane >= first && lane - first < (o_meta & 0x0FFFFFFFu)) {
      const uint8_t v = (uint8_t)in.byte(p);
      const uint32_t at = o_op + (lane - first);
      out.ring[at & out.ring_mask] = v;
      out.dst[at] = v;
    }
  }
  // a literal that goes on behind the window (always the window's last element): what is left of it is moved
  // AFTER the copies, whose ring slots it may take
  uint32_t tail_from = 0, tail_op = 0, tail_len = 0;
  if (T) {
    const uint32_t l_pos = (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)(T - 1u));
    const uint32_t l_meta = (uint32_t)__builtin_amdgcn_readlane((int)e_meta, (int)(T - 1u));
    const uint32_t l_op = (uint32_t)__builtin_amdgcn_readlane((int)e_op, (int)(T - 1u));
    const uint32_t first = l_pos + ((l_meta >> 28) & 7u), len = l_meta & 0x0FFFFFFFu;
    if ((l_meta >> 31) && first + len > 64u) {
      const uint32_t done = first < 64u ? 64u - first : 0u;  // bytes moved above
      tail_from = first < 64u ? 64u : first;                 // window position of the next one
      tail_op = l_op + done;
      tail_len = len - done;
    }
  }
  // ---- copies, in order.  The window's literal bytes are in the ring already, up to output position
  // op + total - tail_len: a copy whose source slots lie a whole ring below a position written above reads
  // its bytes back from memory instead (the slot of source byte s holds s + ring size by now).
  const uint32_t written_end = op + total - tail_len;
  uint64_t copies = __builtin_amdgcn_ballot_w64(mine && !e_lit);
  while (copies) {
    const uint32_t i = (uint32_t)__builtin_ctzll(copies);
    copies &= copies - 1ull;
    const uint32_t c_op = (uint32_t)__builtin_amdgcn_readlane((int)e_op, (int)i);
    const uint32_t c_off = (uint32_t)__builtin_amdgcn_readlane((int)e_off, (int)i);
    out.match(c_op, c_off, (uint32_t)__builtin_amdgcn_readlane((int)e_meta, (int)i) & 0x0FFFFFFFu,
              c_off + (written_end - c_op) > out.ring_mask + 1u);
  }
  if (tail_len) out.literal(in, ip + tail_from, tail_op, tail_len, 1900u);  // (ip + 64 + 1900 lies inside the staged chunks)
  op += total;
  ip = leave == SW_END ? in_end : ip + leave;
  return true;
}

template <int CODEC, uint32_t RING>
__global__ __launch_bounds__(64) void k_lz_pages(const pqd_decompress_desc* __restrict__ descs,
                                                 pqd_page_result* __restrict__ results, uint32_t n_pages) {
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[RING];
  __shared__ __attribute__((aligned(16))) uint8_t s_stage[LZ_STAGE];
  __shared__ uint8_t s_marks[64];
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_decompress_desc d = descs[page];
  if (d.codec != (uint32_t)CODEC) return;  // another kernel's page
  const uint32_t lane = threadIdx.x;
  if (d.src_len > 0x7fffffffu) {
    if (lane == 0) {
      pqd_page_result r = {PQD_BAD_ARG, 0u, 0u, 0u};
      results[page] = r;
    }
    return;
  }
  InStage in;
  in.base = (const PQD_GLOBAL uint8_t*)((uintptr_t)d.src & ~(uintptr_t)15);
  const uint32_t skew = (uint32_t)((uintptr_t)d.src & 15u);
  in.end = skew + d.src_len;
  in.lane = lane;
  in.lds = s_stage;
  in.cur = 0;
  in.start(0);
  in.win_at = 0x80000000u;  // nothing in the window yet
  in.win0 = in.win1 = 0;
  OutRing out = {s_ring, (PQD_GLOBAL uint8_t*)d.dst, lane, RING - 1u};
  const uint32_t cap = d.dst_len, in_end = in.end;
  uint32_t ip = skew, op = 0;
  int32_t status = PQD_OK;

  if (CODEC == PQD_CODEC_SNAPPY) {
    // preamble: the uncompressed length as a varint32
    uint64_t ulen = 0;
    uint32_t shift = 0;
    bool ok = false;
    while (ip < in_end && shift <= 28u) {
      const uint32_t b = (uint32_t)in.peek8(ip) & 0xFFu;
      ++ip;
      ulen |= (uint64_t)(b & 0x7Fu) << shift;
      if (!(b & 0x80u)) {
        ok = true;
        break;
      }
      shift += 7u;
    }
    if (!ok || ulen > 0xFFFFFFFFull) {
      status = PQD_DECODE;
      ip = skew;
    } else {
      bool one_by_one = false;  // set when a window met something out of order: the rest of the page goes this way
      while (ip < in_end) {
        in.reach(ip);
        if (!one_by_one) {
          if (snappy_window(in, out, s_marks, ip, op, in_end, cap, lane)) continue;
          one_by_one = true;
        }
        const uint64_t head = in.peek8(ip);
        const uint32_t tag = (uint32_t)head & 0xFFu;
        const uint32_t rest = (uint32_t)(head >> 8);
        const uint32_t left = in_end - ip;
        const uint32_t kind = tag & 3u;
        if (kind == 0u) {
          uint64_t l = tag >> 2;
          uint32_t hdr = 1;
          if (l >= 60u) {
            const uint32_t nb = (uint32_t)l - 59u;
            if (left < 1u + nb) {
              status = PQD_DECODE;
              break;
            }
            l = nb == 4u ? rest : (rest & ((1u << (8u * nb)) - 1u));
            hdr = 1u + nb;
          }
          const uint64_t len = l + 1u;
          if (len > left - hdr || len > cap - op) {
            status = PQD_DECODE;
            break;
          }
          out.literal(in, ip + hdr, op, (uint32_t)len);
          op += (uint32_t)len;
          ip += hdr + (uint32_t)len;
        } else {
          uint32_t hdr, len, offset;
          if (kind == 1u) {
            hdr = 2;
            len = 4u + ((tag >> 2) & 7u);
            offset = ((tag >> 5) << 8) | (rest & 0xFFu);
          } else if (kind == 2u) {
            hdr = 3;
            len = 1u + (tag >> 2);
            offset = rest & 0xFFFFu;
          } else {
            hdr = 5;
            len = 1u + (tag >> 2);
            offset = rest;
          }
          if (left < hdr || offset == 0u || offset > op || len > cap - op) {
            status = PQD_DECODE;
            break;
          }
          out.match(op, offset, len);
          op += len;
          ip += hdr;
        }
      }
      // the stream produces what its preamble announced, and the page what its header announced
      if (status == PQD_OK && (op != ulen || op != cap)) status = PQD_DECODE;
    }
  } else {  // LZ4 block
    while (ip < in_end) {
      in.reach(ip);
      const uint32_t token = (uint32_t)in.peek8(ip) & 0xFFu;
      ++ip;
      uint64_t lit = token >> 4;
      bool bad = false;
      if (lit == 15u) {
        uint32_t b;
        do {
          if (ip >= in_end) {
            bad = true;
            break;
          }
          in.reach(ip);
          b = (uint32_t)in.peek8(ip) & 0xFFu;
          ++ip;
          lit += b;
        } while (b == 255u);
      }
      if (bad || lit > in_end - ip || lit > cap - op) {
        status = PQD_DECODE;
        break;
      }
      in.reach(ip);
      out.literal(in, ip, op, (uint32_t)lit);
      op += (uint32_t)lit;
      ip += (uint32_t)lit;
      if (ip == in_end) break;  // the last sequence has no match
      if (in_end - ip < 2u) {
        status = PQD_DECODE;
        break;
      }
      in.reach(ip);
      const uint32_t offset = (uint32_t)in.peek8(ip) & 0xFFFFu;
      ip += 2u;
      uint64_t mlen = token & 15u;
      if (mlen == 15u) {
        uint32_t b;
        do {
          if (ip >= in_end) {
            bad = true;
            break;
          }
          in.reach(ip);
          b = (uint32_t)in.peek8(ip) & 0xFFu;
          ++ip;
          mlen += b;
        } while (b == 255u);
      }
      mlen += 4u;
      if (bad || offset == 0u || offset > op || mlen > cap - op) {
        status = PQD_DECODE;
        break;
      }
      out.match(op, offset, (uint32_t)mlen);
      op += (uint32_t)mlen;
    }
    if (status == PQD_OK && op != cap) status = PQD_DECODE;  // the page header announced cap bytes
  }
  if (lane == 0) {
    pqd_page_result r = {status, op, ip - skew, 0u};
    results[page] = r;
  }
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

template <uint32_t RING>
static void launch_lz(const pqd_decompress_desc* descs, pqd_page_result* results, uint32_t n_pages, int codec, hipStream_t stream) {
  if (codec == PQD_CODEC_SNAPPY)
    hipLaunchKernelGGL((k_lz_pages<PQD_CODEC_SNAPPY, RING>), dim3(n_pages), dim3(64), 0, stream, descs, results, n_pages);
  else if (codec == PQD_CODEC_LZ4_RAW)
    hipLaunchKernelGGL((k_lz_pages<PQD_CODEC_LZ4_RAW, RING>), dim3(n_pages), dim3(64), 0, stream, descs, results, n_pages);
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
  const char* v = getenv("PQD_LZ_RING");
  const uint32_t ring = v ? (uint32_t)atoi(v) * 1024u : LZ_RING_DEFAULT;
  if (ring == 8192u)
    launch_lz<8192u>(descs, results, n_pages, codec, stream);
  else if (ring == 32768u)
    launch_lz<32768u>(descs, results, n_pages, codec, stream);
  else
    launch_lz<16384u>(descs, results, n_pages, codec, stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_stored_pages, dim3(n_pages), dim3(256), 0, stream, descs, results, n_pages, (uint32_t)codec);
  return hipGetLastError();
}

}  // namespace pqd
