// Page decompression on the device: raw Snappy and the LZ4 block format (and stored bytes: a copy).
//
// Replaces the step before initFromPage in ColumnChunkPageReadStore.readPage (parquet-hadoop/.../hadoop/
// ColumnChunkPageReadStore.java:151-181 V1, :232-262 V2: decompressor.decompress(bytes, uncompressed size)),
// i.e. SnappyDecompressor.uncompress -> org.xerial.snappy.Snappy.uncompress (codec/SnappyDecompressor.java:
// 27-35) and Lz4RawDecompressor.uncompress -> io.airlift.compress.lz4.Lz4Decompressor (codec/
// Lz4RawDecompressor.java:29-49).  Both formats are a byte-serial stream of "copy these literal bytes" and
// "repeat `length` bytes from `offset` back" elements, so a page is one serial chain; pages are
// independent.  One wave per page: the element headers are decoded wave-uniformly (scalar registers), the
// 64 lanes move the bytes.  The input is staged in LDS in 2 KiB chunks, the next chunk already on its way
// in registers, and the headers are read from a 16-byte window of it kept in scalar registers; the last 32 KiB of output are kept in an LDS ring, which is where nearly every match
// reads from (a match from further back reads the bytes back from memory).  Lane l of a match step takes
// the byte `offset` back, or for offsets below 64 the byte (l mod offset) - offset back of the step's
// start, which is the same byte of the repeating pattern and always already written.
// Bound by the serial element chain, not by HBM: a wave spends several hundred cycles per element (scalar
// decode, branches, one LDS round trip for the bytes), which is about 20 MB/s per page in flight, and LDS
// holds four pages per CU.  (Buffering the output in the ring and writing it in 16-byte pieces, instead of
// storing each element's bytes, was measured and is slower: the stores are not what the chain waits for.)
#include "pqd_device.hpp"
#include "pqd_kernels.hpp"

namespace pqd {

constexpr uint32_t LZ_RING = 32768;
constexpr uint32_t LZ_CHUNK = 2048;
constexpr uint32_t LZ_STAGE = 2 * LZ_CHUNK;

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// (Memory is addressed through global-space pointers: see PQD_GLOBAL.)
struct InStage {
  const PQD_GLOBAL uint8_t* base;  // the source, rounded down to 16 bytes; positions below count from here
  uint32_t end;         // position behind the last source byte
  uint32_t cur;         // chunks cur and cur + 1 are in LDS, chunk cur + 2 in pre0 / pre1
  uint32_t lane;
  uint8_t* lds;
  u32x4 pre0, pre1;

  __device__ __forceinline__ void load(uint32_t c, u32x4& r0, u32x4& r1) const {
    const uint32_t p = c * LZ_CHUNK + lane * 32u;
    r0 = u32x4{0u, 0u, 0u, 0u};
    r1 = r0;
    if (p < end) r0 = *reinterpret_cast<const PQD_GLOBAL u32x4*>(base + p);
    if (p + 16u < end) r1 = *reinterpret_cast<const PQD_GLOBAL u32x4*>(base + p + 16u);
  }
  __device__ __forceinline__ void put(uint32_t c, u32x4 r0, u32x4 r1) {
    u32x4* d = reinterpret_cast<u32x4*>(lds + (c & 1u) * LZ_CHUNK + lane * 32u);
    d[0] = r0;
    d[1] = r1;
  }
  __device__ __forceinline__ void start(uint32_t c) {
    u32x4 a0, a1, b0, b1;
    load(c, a0, a1);
    load(c + 1, b0, b1);
    __syncthreads();
    put(c, a0, a1);
    put(c + 1, b0, b1);
    load(c + 2, pre0, pre1);
    cur = c;
    __syncthreads();
  }
  // makes the chunk of `pos` and the one behind it readable
  __device__ __forceinline__ void reach(uint32_t pos) {
    const uint32_t c = pos / LZ_CHUNK;
    if (c == cur) return;
    if (c == cur + 1) {
      __syncthreads();
      put(cur + 2, pre0, pre1);
      cur = c;
      load(cur + 2, pre0, pre1);
      __syncthreads();
    } else {
      start(c);
    }
  }
  __device__ __forceinline__ uint32_t byte(uint32_t pos) const { return lds[pos & (LZ_STAGE - 1u)]; }
  __device__ __forceinline__ uint32_t ubyte(uint32_t pos) const { return uni(byte(pos)); }

  // The element headers are read from a 16-byte window held in scalar registers: win0 = the eight
  // bytes at the 8-aligned position win_at, win1 the eight behind them.  The window moves on in steps
  // of eight bytes; the half that comes in is requested from LDS when the step is taken and is not
  // needed before the headers have moved past the half in front of it, so its latency is hidden.
  uint64_t win0, win1;
  uint32_t win_at;
  __device__ __forceinline__ uint64_t load8(uint32_t pos8) const {  // pos8: a multiple of 8 inside the staged chunks
    const u32x2 v = *reinterpret_cast<const u32x2*>(lds + (pos8 & (LZ_STAGE - 1u)));
    return ((uint64_t)uni(v.y) << 32) | uni(v.x);
  }
  // the eight bytes at `pos` (reach(pos) has been called; pos + 8 may lie one chunk further)
  __device__ __forceinline__ uint64_t peek8(uint32_t pos) {
    if (pos - win_at >= 16u) {  // a jump (long literal, first use): both halves anew
      win_at = pos & ~7u;
      win0 = load8(win_at);
      win1 = load8(win_at + 8u);
    } else if (pos - win_at >= 8u) {
      win_at += 8u;
      win0 = win1;
      win1 = load8(win_at + 8u);
    }
    const uint32_t sh = (pos - win_at) * 8u;
    return sh ? (win0 >> sh) | (win1 << (64u - sh)) : win0;
  }
};

struct OutRing {
  uint8_t* ring;
  PQD_GLOBAL uint8_t* dst;
  uint32_t lane;

  // `len` literal bytes from the staged input (they lie within the staged chunks) or, long ones, from memory
  __device__ __forceinline__ void literal(InStage& in, uint32_t from, uint32_t op, uint32_t len) {
    if (len <= LZ_CHUNK - 16u) {
      for (uint32_t i = lane; i < len; i += 64u) {
        const uint8_t v = (uint8_t)in.byte(from + i);
        ring[(op + i) & (LZ_RING - 1u)] = v;
        dst[op + i] = v;
      }
    } else {
      const PQD_GLOBAL uint8_t* src = in.base + from;
#pragma unroll 4
      for (uint32_t i = lane; i < len; i += 64u) {
        const uint8_t v = src[i];
        ring[(op + i) & (LZ_RING - 1u)] = v;
        dst[op + i] = v;
      }
    }
  }
  __device__ __forceinline__ void match(uint32_t op, uint32_t offset, uint32_t len) {
    uint32_t lm = lane;  // lane mod offset for offsets below 64: six conditional subtractions
    if (offset < 64u) {
#pragma unroll
      for (int k = 5; k >= 0; --k) {
        const uint32_t t = offset << k;
        lm = lm >= t ? lm - t : lm;
      }
    }
    if (offset <= LZ_RING) {
      for (uint32_t b = 0; b < len; b += 64u) {
        if (b + lane < len) {
          const uint8_t v = ring[(op + b + lm - offset) & (LZ_RING - 1u)];
          ring[(op + b + lane) & (LZ_RING - 1u)] = v;
          dst[op + b + lane] = v;
        }
      }
    } else {
      for (uint32_t b = 0; b < len; b += 64u) {
        __threadfence();  // the bytes were stored by this wave: read them where the stores went
        if (b + lane < len) {
          const uint8_t v = __hip_atomic_load(dst + (op + b + lane - offset), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ring[(op + b + lane) & (LZ_RING - 1u)] = v;
          dst[op + b + lane] = v;
        }
      }
    }
  }
};

template <int CODEC>
__global__ __launch_bounds__(64) void k_lz_pages(const pqd_decompress_desc* __restrict__ descs,
                                                 pqd_page_result* __restrict__ results, uint32_t n_pages) {
  __shared__ __attribute__((aligned(16))) uint8_t s_ring[LZ_RING];
  __shared__ __attribute__((aligned(16))) uint8_t s_stage[LZ_STAGE];
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
  OutRing out = {s_ring, (PQD_GLOBAL uint8_t*)d.dst, lane};
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
      while (ip < in_end) {
        in.reach(ip);
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
__global__ __launch_bounds__(256) void k_stored_pages(const pqd_decompress_desc* __restrict__ descs,
                                                      pqd_page_result* __restrict__ results, uint32_t n_pages) {
  const uint32_t page = blockIdx.x;
  if (page >= n_pages) return;
  const pqd_decompress_desc d = descs[page];
  if (d.codec != PQD_CODEC_UNCOMPRESSED) return;
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

hipError_t launch_decompress_pages(const pqd_decompress_desc* descs, pqd_page_result* results, uint32_t n_pages, int codec,
                                   hipStream_t stream) {
  if (n_pages == 0) return hipSuccess;
  if (codec == PQD_CODEC_SNAPPY)
    hipLaunchKernelGGL(k_lz_pages<PQD_CODEC_SNAPPY>, dim3(n_pages), dim3(64), 0, stream, descs, results, n_pages);
  else if (codec == PQD_CODEC_LZ4_RAW)
    hipLaunchKernelGGL(k_lz_pages<PQD_CODEC_LZ4_RAW>, dim3(n_pages), dim3(64), 0, stream, descs, results, n_pages);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_stored_pages, dim3(n_pages), dim3(256), 0, stream, descs, results, n_pages);
  return hipGetLastError();
}

}  // namespace pqd
