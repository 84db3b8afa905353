// Column chunks: the page-header walk (host) and the cut of every data page into its level and value
// sections with their decode descriptors (device).
//
// The walk replaces ParquetFileReader.Chunk.readAllPages (parquet-hadoop/.../hadoop/
// ParquetFileReader.java:1973-2117): PageHeader structs in thrift's compact protocol (parquet-format
// PageHeader, DataPageHeader, DictionaryPageHeader, DataPageHeaderV2; read in the reference through
// parquet-format-structures' Util.readPageHeader), one after the other with the page bytes between
// them, until the chunk's value count is reached; optional CRC-32 check of the page bytes (:1954-1962).
// The section cut replaces ColumnReaderBase.readPageV1 / readPageV2 / initDataReader
// (parquet-column/.../impl/ColumnReaderBase.java:693-780) for all pages of the chunk in one launch.
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/pqd.h"
#include "pqd_kernels.hpp"

namespace {

// ---- thrift compact protocol, as much of it as a PageHeader needs --------------------------------
struct Reader {
  const uint8_t* p;
  size_t len, pos;
  bool eof = false, bad = false;

  uint32_t byte() {
    if (pos >= len) {
      eof = true;
      return 0;
    }
    return p[pos++];
  }
  uint64_t varint() {
    uint64_t v = 0;
    for (uint32_t shift = 0; shift < 64; shift += 7) {
      const uint32_t b = byte();
      if (eof) return 0;
      v |= (uint64_t)(b & 0x7Fu) << shift;
      if (!(b & 0x80u)) return v;
    }
    bad = true;  // more than ten bytes
    return 0;
  }
  int64_t zigzag() {
    const uint64_t v = varint();
    return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
  }
  void skip_bytes(uint64_t n) {
    if (n > len - pos) {
      eof = true;
      pos = len;
    } else {
      pos += (size_t)n;
    }
  }
  // compact-protocol type codes: 1/2 bool, 3 byte, 4 i16, 5 i32, 6 i64, 7 double, 8 binary, 9 list, 10 set, 11 map, 12 struct
  void skip(uint32_t type, int depth) {
    if (depth > 16 || eof || bad) {
      bad = bad || depth > 16;
      return;
    }
    switch (type) {
      case 1:
      case 2:
        return;
      case 3:
        (void)byte();
        return;
      case 4:
      case 5:
      case 6:
        (void)varint();
        return;
      case 7:
        skip_bytes(8);
        return;
      case 8:
        skip_bytes(varint());
        return;
      case 9:
      case 10: {
        const uint32_t h = byte();
        uint64_t n = h >> 4;
        if (n == 15) n = varint();
        const uint32_t et = h & 15u;
        for (uint64_t i = 0; i < n && !eof && !bad; ++i) {
          if (et == 1 || et == 2)
            (void)byte();  // a bool inside a collection is a byte of its own
          else
            skip(et, depth + 1);
        }
        return;
      }
      case 11: {
        const uint64_t n = varint();
        if (n == 0) return;
        const uint32_t kv = byte();
        for (uint64_t i = 0; i < n && !eof && !bad; ++i) {
          for (uint32_t t : {kv >> 4, kv & 15u}) {
            if (t == 1 || t == 2)
              (void)byte();
            else
              skip(t, depth + 1);
          }
        }
        return;
      }
      case 12:
        skip_struct(depth + 1);
        return;
      default:
        bad = true;
    }
  }
  // field header: returns false at the struct's stop byte
  bool field(int32_t& id, uint32_t& type) {
    const uint32_t h = byte();
    if (eof || h == 0) return false;
    type = h & 15u;
    const uint32_t delta = h >> 4;
    if (delta)
      id += (int32_t)delta;
    else
      id = (int32_t)zigzag();
    return !eof && !bad;
  }
  void skip_struct(int depth) {
    int32_t id = 0;
    uint32_t type;
    while (field(id, type)) skip(type, depth);
  }
};

struct Header {
  int64_t f[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // small integer fields by id (1-based)
  bool set[9] = {false, false, false, false, false, false, false, false, false};
  bool flag[9] = {false, false, false, false, false, false, false, false, false};
};

// Integer and bool fields 1..8 of a struct; everything else is skipped.
void read_small_struct(Reader& r, Header& h) {
  int32_t id = 0;
  uint32_t type;
  while (r.field(id, type)) {
    if (id >= 1 && id <= 8 && (type == 1 || type == 2)) {
      h.flag[id] = type == 1;
      h.set[id] = true;
    } else if (id >= 1 && id <= 8 && type >= 3 && type <= 6) {
      h.f[id] = type == 3 ? (int64_t)(int8_t)r.byte() : r.zigzag();
      h.set[id] = true;
    } else {
      r.skip(type, 0);
    }
  }
}

uint32_t g_crc_table[8][256];

void crc_init() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    g_crc_table[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_crc_table[t][i] = (g_crc_table[t - 1][i] >> 8) ^ g_crc_table[0][g_crc_table[t - 1][i] & 0xFFu];
}

}  // namespace

extern "C" {

uint32_t pqd_crc32(uint32_t crc, const uint8_t* data, size_t len) {
  static const bool ready = (crc_init(), true);
  (void)ready;
  uint32_t c = ~crc;
  while (len >= 8) {  // eight bytes per step
    uint32_t lo, hi;
    memcpy(&lo, data, 4);
    memcpy(&hi, data + 4, 4);
    lo ^= c;
    c = g_crc_table[7][lo & 0xFFu] ^ g_crc_table[6][(lo >> 8) & 0xFFu] ^ g_crc_table[5][(lo >> 16) & 0xFFu] ^
        g_crc_table[4][lo >> 24] ^ g_crc_table[3][hi & 0xFFu] ^ g_crc_table[2][(hi >> 8) & 0xFFu] ^
        g_crc_table[1][(hi >> 16) & 0xFFu] ^ g_crc_table[0][hi >> 24];
    data += 8;
    len -= 8;
  }
  while (len--) c = g_crc_table[0][(c ^ *data++) & 0xFFu] ^ (c >> 8);
  return ~c;
}

int pqd_chunk_walk(const uint8_t* chunk, size_t chunk_len, int64_t value_count, int verify_crc, pqd_chunk_page* pages,
                   uint32_t pages_capacity, uint32_t* n_pages) {
  if (!n_pages || (chunk_len && !chunk) || (pages_capacity && !pages)) return PQD_BAD_ARG;
  *n_pages = 0;
  if (chunk_len > 0xFFFFFFFFull) return PQD_BAD_ARG;  // offsets are 32-bit; a chunk is far smaller
  Reader r{chunk, chunk_len, 0};
  uint32_t count = 0;
  int64_t values_read = 0;
  bool have_dictionary = false;
  // hasMorePages (:2130-2134): by value count, or until the bytes are used up
  while (value_count >= 0 ? values_read < value_count : r.pos < chunk_len) {
    pqd_chunk_page pg;
    memset(&pg, 0, sizeof pg);
    pg.header_offset = (uint32_t)r.pos;
    Header top, sub;
    int sub_of = 0;  // which page-header struct was present: 5, 7 or 8
    {
      int32_t id = 0;
      uint32_t type;
      while (r.field(id, type)) {
        if (id >= 1 && id <= 4 && type >= 3 && type <= 6) {
          top.f[id] = type == 3 ? (int64_t)(int8_t)r.byte() : r.zigzag();
          top.set[id] = true;
        } else if ((id == 5 || id == 7 || id == 8) && type == 12) {
          Header h;
          read_small_struct(r, h);
          // the struct that matches the page type counts (a header could carry others)
          if (!sub_of || id == (top.f[1] == PQD_PAGE_DATA ? 5 : top.f[1] == PQD_PAGE_DICTIONARY ? 7 : 8)) {
            sub = h;
            sub_of = id;
          }
        } else {
          r.skip(type, 0);
        }
      }
    }
    if (r.eof) return PQD_EOF;
    if (r.bad || !top.set[1] || !top.set[2] || !top.set[3]) return PQD_DECODE;  // required fields (thrift: TProtocolException)
    const int64_t uncompressed = top.f[2], compressed = top.f[3];
    if (uncompressed < 0 || compressed < 0 || uncompressed > 0x7FFFFFFF || compressed > 0x7FFFFFFF) return PQD_DECODE;
    pg.body_offset = (uint32_t)r.pos;
    pg.compressed_size = (uint32_t)compressed;
    pg.uncompressed_size = (uint32_t)uncompressed;
    pg.has_crc = top.set[4];
    pg.crc = (uint32_t)top.f[4];
    pg.type = (uint8_t)top.f[1];
    pg.is_compressed = 1;
    if ((uint64_t)compressed > chunk_len - r.pos) return PQD_EOF;  // readAsBytesInput past the chunk
    const int64_t type = top.f[1];
    bool keep = true;
    if (type == PQD_PAGE_DICTIONARY) {
      if (have_dictionary) return PQD_DECODE;  // "more than one dictionary page in column"
      if (sub_of != 7 || !sub.set[1] || !sub.set[2] || sub.f[1] < 0) return PQD_DECODE;
      have_dictionary = true;
      pg.num_values = (uint32_t)sub.f[1];
      pg.encoding = (uint8_t)sub.f[2];
    } else if (type == PQD_PAGE_DATA) {
      if (sub_of != 5 || !sub.set[1] || !sub.set[2] || !sub.set[3] || !sub.set[4] || sub.f[1] < 0) return PQD_DECODE;
      pg.num_values = (uint32_t)sub.f[1];
      pg.encoding = (uint8_t)sub.f[2];
      pg.def_encoding = (uint8_t)sub.f[3];
      pg.rep_encoding = (uint8_t)sub.f[4];
      values_read += sub.f[1];
    } else if (type == PQD_PAGE_DATA_V2) {
      if (sub_of != 8 || !sub.set[1] || !sub.set[2] || !sub.set[3] || !sub.set[4] || !sub.set[5] || !sub.set[6] || sub.f[1] < 0)
        return PQD_DECODE;
      // dataSize = compressedPageSize - repetition_levels_byte_length - definition_levels_byte_length (:2063-2065)
      if (sub.f[5] < 0 || sub.f[6] < 0 || sub.f[5] + sub.f[6] > compressed) return PQD_DECODE;
      pg.num_values = (uint32_t)sub.f[1];
      pg.num_nulls = (uint32_t)sub.f[2];
      pg.num_rows = (uint32_t)sub.f[3];
      pg.encoding = (uint8_t)sub.f[4];
      pg.def_levels_len = (uint32_t)sub.f[5];
      pg.rep_levels_len = (uint32_t)sub.f[6];
      pg.def_encoding = pg.rep_encoding = PQD_FMT_RLE;
      pg.is_compressed = sub.set[7] ? (sub.flag[7] ? 1 : 0) : 1;
      values_read += sub.f[1];
    } else {
      keep = false;  // "skipping page of type ..." (:2098-2101)
    }
    if (keep && verify_crc && pg.has_crc && pqd_crc32(0, chunk + r.pos, (size_t)compressed) != pg.crc)
      return PQD_DECODE;  // "could not verify page integrity, CRC checksum verification failed"
    r.pos += (size_t)compressed;
    if (keep) {
      if (count < pages_capacity) pages[count] = pg;
      ++count;
    }
  }
  *n_pages = count;
  if (value_count >= 0 && values_read != value_count) return PQD_DECODE;  // "Expected N values in column chunk ... but got M"
  return count > pages_capacity ? PQD_BAD_ARG : PQD_OK;
}

}  // extern "C"

// ---- device: sections and descriptors -----------------------------------------------------------------
namespace pqd {

__device__ __forceinline__ uint32_t level_width(int32_t max_level) {  // BytesUtils.getWidthFromMaxInt (BytesUtils.java:49-51)
  return max_level <= 0 ? 0u : 32u - (uint32_t)__builtin_clz((uint32_t)max_level);
}

__global__ __launch_bounds__(256) void k_chunk_sections(const pqd_chunk_page* __restrict__ sections, uint32_t n_pages,
                                                        pqd_column_meta col, pqd_chunk_outputs o,
                                                        pqd_page_desc* __restrict__ rep_descs, pqd_page_desc* __restrict__ def_descs,
                                                        pqd_page_desc* __restrict__ val_descs, pqd_page_result* __restrict__ results) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_pages) return;
  const pqd_chunk_page pg = sections[i];
  const uint8_t* body = o.bodies + pg.body_offset;
  const uint32_t size = pg.uncompressed_size;
  const uint32_t n = pg.num_values;
  const uint32_t lbase = o.level_base ? o.level_base[i] : 0u;
  // The records describe UNCOMPRESSED bodies (the walk's records of an uncompressed column, or the records
  // pqd_chunk_decompress_dev returns): a record whose two sizes differ has only compressed_size bytes behind it.
  int32_t status = pg.compressed_size == pg.uncompressed_size ? PQD_OK : PQD_DECODE;
  uint32_t pos = 0;
  pqd_page_desc lv[2];
  // repetition levels first, then definition levels (readPageV1 :734-744, readPageV2 :770-777)
  for (int k = 0; k < 2; ++k) {
    const int32_t max_level = k == 0 ? col.max_rep : col.max_def;
    const uint32_t enc = k == 0 ? pg.rep_encoding : pg.def_encoding;
    pqd_page_desc d = {};
    d.out = k == 0 ? (void*)(o.rep_out ? o.rep_out + lbase : nullptr) : (void*)(o.def_out ? o.def_out + lbase : nullptr);
    d.num_values = n;
    d.elem_size = 4;
    d.encoding = PQD_ENC_RLE;
    d.bit_width = (uint8_t)level_width(max_level);
    uint32_t len = 0;
    if (max_level > 0 && status == PQD_OK) {
      if (pg.type == PQD_PAGE_DATA_V2) {
        len = k == 0 ? pg.rep_levels_len : pg.def_levels_len;  // no prefix: the lengths come from the header
        if (len > size - pos) status = PQD_DECODE;
      } else if (enc == PQD_FMT_BIT_PACKED) {
        d.encoding = PQD_ENC_BIT_PACKED;
        const uint64_t want = ((uint64_t)n * d.bit_width + 7) >> 3;  // ByteBitPackingValuesReader.java:78-85
        len = want > size - pos ? size - pos : (uint32_t)want;
      } else if (enc == PQD_FMT_RLE) {
        d.flags = PQD_F_LENGTH_PREFIX;
        if (size - pos < 4) {
          status = PQD_EOF;
        } else {
          const uint32_t l = (uint32_t)body[pos] | ((uint32_t)body[pos + 1] << 8) | ((uint32_t)body[pos + 2] << 16) |
                             ((uint32_t)body[pos + 3] << 24);
          if ((int32_t)l < 0)
            status = PQD_DECODE;
          else if (l > size - pos - 4)
            status = PQD_EOF;
          else
            len = 4 + l;
        }
      } else {
        status = PQD_DECODE;  // no other level encoding exists (Encoding.java: only RLE and BIT_PACKED read levels)
      }
    }
    if (status != PQD_OK) len = 0;
    d.data = body + pos;
    d.data_len = len;
    if (max_level <= 0 || d.out == nullptr) d.num_values = 0;  // nothing to decode: all zero
    pos += len;
    lv[k] = d;
  }
  rep_descs[i] = lv[0];
  def_descs[i] = lv[1];
  // the values: what is left of the page (initDataReader :693-727)
  pqd_page_desc v = {};
  uint32_t elem = 0;
  switch (col.physical_type) {
    case PQD_TYPE_INT32:
    case PQD_TYPE_FLOAT:
      elem = 4;
      break;
    case PQD_TYPE_INT64:
    case PQD_TYPE_DOUBLE:
      elem = 8;
      break;
    case PQD_TYPE_INT96:
      elem = 12;
      break;
    case PQD_TYPE_FIXED_LEN_BYTE_ARRAY:
      elem = col.type_length > 0 && col.type_length <= 16 ? (uint32_t)col.type_length : 0u;
      break;
    default:
      break;
  }
  const bool fixed = elem != 0;
  uint32_t enc = 0, out_elem = elem;
  switch (pg.encoding) {
    case PQD_FMT_PLAIN:
      if (col.physical_type == PQD_TYPE_BOOLEAN) {
        enc = PQD_ENC_PLAIN_BOOLEAN;
        out_elem = 1;
      } else if (fixed) {
        enc = PQD_ENC_PLAIN;
      } else if (col.physical_type == PQD_TYPE_BYTE_ARRAY) {
        enc = PQD_ENC_PLAIN_BYTE_ARRAY;  // Encoding.PLAIN.getValuesReader: BinaryPlainValuesReader (Encoding.java:63-86)
        out_elem = 0;
      }
      break;
    case PQD_FMT_RLE:  // booleans: RunLengthBitPackingHybridValuesReader(1), length-prefixed (Encoding.java:119-145)
      if (col.physical_type == PQD_TYPE_BOOLEAN) {
        enc = PQD_ENC_RLE;
        out_elem = 4;
        v.bit_width = 1;
        v.flags = PQD_F_LENGTH_PREFIX;
      }
      break;
    case PQD_FMT_PLAIN_DICTIONARY:
    case PQD_FMT_RLE_DICTIONARY:
      // fixed-width dictionaries gather 4- or 8-byte values; BYTE_ARRAY kinds gather 8-byte views
      if (elem == 4 || elem == 8) {
        enc = PQD_ENC_RLE_DICTIONARY;
      } else if (col.physical_type == PQD_TYPE_BYTE_ARRAY || col.physical_type == PQD_TYPE_FIXED_LEN_BYTE_ARRAY ||
                 col.physical_type == PQD_TYPE_INT96) {
        enc = PQD_ENC_RLE_DICTIONARY;
        out_elem = 8;
      }
      break;
    case PQD_FMT_DELTA_BINARY_PACKED:
      if (col.physical_type == PQD_TYPE_INT32 || col.physical_type == PQD_TYPE_INT64) enc = PQD_ENC_DELTA_BINARY_PACKED;
      break;
    case PQD_FMT_BYTE_STREAM_SPLIT:
      if (fixed) enc = PQD_ENC_BYTE_STREAM_SPLIT;
      break;
    case PQD_FMT_DELTA_LENGTH_BYTE_ARRAY:  // Encoding.java:204-217: BYTE_ARRAY only
      if (col.physical_type == PQD_TYPE_BYTE_ARRAY) {
        enc = PQD_ENC_DELTA_LENGTH_BYTE_ARRAY;
        out_elem = 0;
      }
      break;
    case PQD_FMT_DELTA_BYTE_ARRAY:  // Encoding.java:219-232: BYTE_ARRAY and FIXED_LEN_BYTE_ARRAY
      if (col.physical_type == PQD_TYPE_BYTE_ARRAY || col.physical_type == PQD_TYPE_FIXED_LEN_BYTE_ARRAY) {
        enc = PQD_ENC_DELTA_BYTE_ARRAY;
        out_elem = 0;
      }
      break;
    default:
      break;
  }
  // (the three BYTE_ARRAY encodings: elem_size 0, `out` not set; the caller makes a pqd_delta_bytes_desc of data,
  // data_len, num_values and encoding)
  v.data = body + pos;
  v.data_len = status == PQD_OK ? size - pos : 0u;
  v.num_values = n;
  v.encoding = (uint8_t)enc;
  v.elem_size = (uint8_t)out_elem;
  v.out = o.val_out && out_elem ? (uint8_t*)o.val_out + (size_t)(o.value_base ? o.value_base[i] : lbase) * out_elem : nullptr;
  if (enc == PQD_ENC_RLE_DICTIONARY) {
    v.dict = o.dict;
    v.dict_len = o.dict_len;
  }
  val_descs[i] = v;
  pqd_page_result r = {status, n, pos, v.data_len};
  results[i] = r;
}

__global__ __launch_bounds__(256) void k_chunk_set_value_counts(pqd_page_desc* __restrict__ val_descs,
                                                                const pqd_page_result* __restrict__ validity, uint32_t n_pages) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n_pages) val_descs[i].num_values = validity[i].reserved;
}

}  // namespace pqd

extern "C" {

int pqd_chunk_sections_dev(const pqd_chunk_page* sections, uint32_t n_pages, const pqd_column_meta* column,
                           const pqd_chunk_outputs* outputs, pqd_page_desc* rep_descs, pqd_page_desc* def_descs,
                           pqd_page_desc* val_descs, pqd_page_result* results, void* stream) {
  if (n_pages == 0) return PQD_OK;
  if (!sections || !column || !outputs || !rep_descs || !def_descs || !val_descs || !results || !outputs->bodies) return PQD_BAD_ARG;
  hipLaunchKernelGGL(pqd::k_chunk_sections, dim3((n_pages + 255u) / 256u), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     sections, n_pages, *column, *outputs, rep_descs, def_descs, val_descs, results);
  return hipGetLastError() == hipSuccess ? PQD_OK : PQD_HIP_ERROR;
}

int pqd_chunk_set_value_counts_dev(pqd_page_desc* val_descs, const pqd_page_result* validity_results, uint32_t n_pages,
                                   void* stream) {
  if (n_pages == 0) return PQD_OK;
  if (!val_descs || !validity_results) return PQD_BAD_ARG;
  hipLaunchKernelGGL(pqd::k_chunk_set_value_counts, dim3((n_pages + 255u) / 256u), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), val_descs, validity_results, n_pages);
  return hipGetLastError() == hipSuccess ? PQD_OK : PQD_HIP_ERROR;
}

}  // extern "C"
