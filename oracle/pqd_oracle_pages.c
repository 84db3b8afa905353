/*
 * pqd_oracle_pages.c -- CPU restatement of the steps on either side of the value decoders: PLAIN
 * pages, and what ColumnReaderBase does with decoded levels.
 *
 * TEST INFRASTRUCTURE ONLY (see pqd_oracle.h).  Each function cites the reference file it restates
 * (paths relative to the apache/parquet-java tree) and is written the way the Java is: one value
 * at a time.
 */
#include <stdlib.h>
#include <string.h>

#include "pqd_oracle.h"

/* PlainValuesReader.{Integer,Long,Float,Double}PlainValuesReader and FixedLenByteArrayPlainValuesReader
 * (parquet-column/.../values/plain/PlainValuesReader.java:52-60: initFromPage slices everything that is
 * available; :62-145: each read takes the next little-endian value, BufferUnderflowException ->
 * ParquetDecodingException("could not read ...")).  `n` reads of `elem` bytes each. */
int orc_plain_decode(const uint8_t* in, size_t in_len, int elem, size_t n, uint8_t* out, size_t* written) {
  size_t pos = 0;
  *written = 0;
  if (elem < 1 || elem > 16) return ORC_BAD_ARG;
  for (size_t i = 0; i < n; ++i) {
    if (in_len - pos < (size_t)elem) return ORC_DECODE; /* buffer.getInt() past the limit */
    memcpy(out + i * (size_t)elem, in + pos, (size_t)elem);
    pos += (size_t)elem;
    *written = i + 1;
  }
  return ORC_OK;
}

/* BooleanPlainValuesReader (parquet-column/.../values/plain/BooleanPlainValuesReader.java:48-78):
 * length = min(paddedByteCountFromBits(valueCount), available); readBoolean fails once
 * bitIndex >= length * 8; value = (pageData[bitIndex >>> 3] >>> (bitIndex & 7)) & 1. */
int orc_plain_boolean_decode(const uint8_t* in, size_t in_len, size_t n, uint8_t* out, size_t* written,
                             size_t* consumed) {
  size_t length = (n + 7) / 8;
  if (length > in_len) length = in_len;
  const size_t bit_count = length * 8;
  *consumed = length; /* updateNextOffset(length) */
  *written = 0;
  for (size_t bit_index = 0; bit_index < n; ++bit_index) {
    if (bit_index >= bit_count) return ORC_DECODE; /* "attempt to read beyond end of boolean page" */
    out[bit_index] = (uint8_t)((in[bit_index >> 3] >> (bit_index & 7)) & 1);
    *written = bit_index + 1;
  }
  return ORC_OK;
}

/* What ColumnReaderBase knows per value (parquet-column/.../impl/ColumnReaderBase.java:642-668 checkRead,
 * :322-325 isNull): a value is present exactly when its definition level equals the column's maximum.
 * validity bit i (LSB first), value_index[i] = how many values came before it, -1 for a null. */
void orc_def_validity(const int32_t* def, size_t n, int32_t max_def, uint8_t* validity, int32_t* value_index,
                      size_t* non_null) {
  size_t read_values = 0;
  if (validity) memset(validity, 0, (n + 7) / 8);
  for (size_t i = 0; i < n; ++i) {
    const int32_t dl = def ? def[i] : max_def; /* max level 0: no level bytes, every value present */
    const int present = dl == max_def;
    if (present && validity) validity[i >> 3] |= (uint8_t)(1u << (i & 7));
    if (value_index) value_index[i] = present ? (int32_t)read_values : -1;
    read_values += (size_t)present;
  }
  *non_null = read_values;
}

/* The same loop, handing the values out: slot i gets the next value of the page when it is present
 * (binding.read() in ColumnReaderBase.java:576-581 is only reached for present values), zero bytes
 * otherwise.  A present slot beyond the page's values: the Java read fails there. */
int orc_scatter_nulls(const uint8_t* values, size_t n_values, const uint8_t* validity, size_t n_slots, int elem,
                      uint8_t* out, size_t* written, size_t* set_bits) {
  size_t next = 0;
  int status = ORC_OK;
  *written = n_slots;
  for (size_t i = 0; i < n_slots; ++i) {
    const int present = validity ? (validity[i >> 3] >> (i & 7)) & 1 : 1;
    uint8_t* o = out + i * (size_t)elem;
    if (present && next < n_values) {
      memcpy(o, values + next * (size_t)elem, (size_t)elem);
    } else {
      memset(o, 0, (size_t)elem);
      if (present && status == ORC_OK) {
        status = ORC_DECODE;
        *written = i;
      }
    }
    next += (size_t)present;
  }
  *set_bits = next;
  return status;
}

/* Records and their element slots from the level pairs: a record starts where the repetition level is 0
 * (ColumnReaderBase.java:664-666 keeps rl; RecordReaderImplementation.java:409-429 ends a record when the
 * next repetition level is 0); a level pair stands for an element slot of the outermost list when its
 * definition level reaches slot_def, and the list itself is defined when the record's first definition
 * level reaches row_def. */
int orc_rep_offsets(const int32_t* rep, const int32_t* def, size_t n, int32_t slot_def, int32_t row_def, size_t rows_cap,
                    int32_t* offsets, uint8_t* row_valid, size_t* n_rows, size_t* n_slots) {
  size_t rows = 0, slots = 0;
  for (size_t i = 0; i < n; ++i) {
    if (rep[i] == 0) {
      if (rows < rows_cap) {
        offsets[rows] = (int32_t)slots;
        if (row_valid) row_valid[rows] = def[i] >= row_def;
      }
      ++rows;
    }
    if (def[i] >= slot_def) ++slots;
  }
  if (rows <= rows_cap) offsets[rows] = (int32_t)slots;
  *n_rows = rows;
  *n_slots = slots;
  return rows <= rows_cap ? ORC_OK : ORC_BAD_ARG;
}

/* ------------------------------------------------------------------------- */
/* DELTA_LENGTH_BYTE_ARRAY, DELTA_BYTE_ARRAY                                    */
/* ------------------------------------------------------------------------- */

/* One DELTA_BINARY_PACKED block of lengths read as ints: readInteger is (int) readLong
 * (DeltaBinaryPackingValuesReader.java:96-100).  `cap` bounds the header's count as the value readers
 * of this library do (the page cannot hold more lengths than it has values). */
static int read_length_block(const uint8_t* in, size_t in_len, size_t cap, int32_t** lengths, size_t* count,
                             size_t* consumed) {
  int64_t* wide = (int64_t*)calloc(cap + 1, sizeof(int64_t));
  if (!wide) return ORC_DECODE;
  size_t total = 0;
  int st = orc_delta_decode(in, in_len, wide, cap, &total, consumed);
  if (st != ORC_OK) {
    free(wide);
    return st;
  }
  int32_t* narrow = (int32_t*)calloc(total + 1, sizeof(int32_t));
  if (!narrow) {
    free(wide);
    return ORC_DECODE;
  }
  for (size_t i = 0; i < total; ++i) narrow[i] = (int32_t)wide[i];
  free(wide);
  *lengths = narrow;
  *count = total;
  return ORC_OK;
}

/* BinaryPlainValuesReader.java:36-74 (parquet-column/.../values/plain/): initFromPage keeps everything that is
 * available; readBytes reads the length (BytesUtils.readIntLittleEndian: four bytes, EOFException when fewer are
 * left) and slices that many bytes (EOFException when the page ends first); either failure becomes
 * ParquetDecodingException("could not read bytes at offset ...").  A negative length is reported as a decode
 * error (the reference's streams disagree on it, as for the delta encodings).  offsets[i] = sum of the lengths in
 * front of value i; bytes may be NULL (the values stay in the page: value i at in[offsets[i] + 4 * (i + 1)]).
 * *consumed = the page bytes in front of the first value not delivered. */
int orc_plain_byte_array(const uint8_t* in, size_t in_len, size_t n, int32_t* offsets, uint8_t* bytes, size_t cap,
                         size_t* written, size_t* consumed, size_t* bytes_written) {
  *written = *consumed = *bytes_written = 0;
  offsets[0] = 0;
  if (in_len > 0x7fffffffu) return ORC_BAD_ARG;
  if (cap > 0x7fffffffu || !bytes) cap = 0x7fffffffu;
  int st = ORC_OK;
  size_t pos = 0, out = 0, i = 0;
  for (; i < n; ++i) {
    offsets[i] = (int32_t)out;
    if (in_len - pos < 4) {
      st = ORC_DECODE;
      break;
    }
    const uint32_t raw = (uint32_t)in[pos] | ((uint32_t)in[pos + 1] << 8) | ((uint32_t)in[pos + 2] << 16) | ((uint32_t)in[pos + 3] << 24);
    if ((int32_t)raw < 0 || (size_t)raw > in_len - pos - 4) {
      st = ORC_DECODE;
      break;
    }
    if (out + raw > cap) {
      st = ORC_BAD_ARG;
      break;
    }
    if (bytes) memcpy(bytes + out, in + pos + 4, raw);
    pos += 4 + (size_t)raw;
    out += raw;
  }
  offsets[i] = (int32_t)out;
  *written = i;
  *consumed = pos;
  *bytes_written = out;
  return st;
}

/* DeltaLengthByteArrayValuesReader.java:44-58: initFromPage reads every length, the values follow back
 * to back; readBytes takes the next length and slices that many bytes ("Failed to read N bytes" when
 * the page ends first).  A length block with fewer lengths than the page has values fails the read
 * behind its last length ("no more value to read", DeltaBinaryPackingValuesReader.java:102-107).  A
 * negative length is reported as a decode error (the reference's streams disagree on it). */
int orc_delta_length_byte_array(const uint8_t* in, size_t in_len, size_t n, int32_t* offsets, uint8_t* bytes,
                                size_t cap, size_t* written, size_t* consumed, size_t* bytes_written) {
  *written = *consumed = *bytes_written = 0;
  offsets[0] = 0;
  int32_t* lengths = NULL;
  size_t count = 0, header = 0;
  int st = read_length_block(in, in_len, n, &lengths, &count, &header);
  if (st != ORC_OK) return st;
  if (cap > 0x7fffffffu) cap = 0x7fffffffu;
  size_t pos = header, out = 0, i = 0;
  for (; i < n; ++i) {
    offsets[i] = (int32_t)out;
    if (i >= count) {
      st = ORC_DECODE;
      break;
    }
    int32_t len = lengths[i];
    if (len < 0 || (size_t)len > in_len - pos) {
      st = ORC_DECODE;
      break;
    }
    if (out + (size_t)len > cap) {
      st = ORC_BAD_ARG;
      break;
    }
    if (bytes) memcpy(bytes + out, in + pos, (size_t)len);
    pos += (size_t)len;
    out += (size_t)len;
  }
  offsets[i] = (int32_t)out;
  free(lengths);
  *written = i;
  *consumed = pos;
  *bytes_written = out;
  return st;
}

/* DeltaByteArrayReader.java:50-85: initFromPage reads the prefix lengths, then hands the rest of the
 * page to a DeltaLengthByteArrayValuesReader for the suffixes; readBytes: prefixLength, suffix, and
 * when prefixLength != 0 the value is previous[0, prefixLength) + suffix (System.arraycopy: a prefix
 * longer than the previous value is an ArrayIndexOutOfBoundsException), else the suffix itself.
 * `prev` is the last value of the page before (setPreviousReader, :87-95), or empty. */
int orc_delta_byte_array(const uint8_t* in, size_t in_len, size_t n, const uint8_t* prev, size_t prev_len,
                         int32_t* offsets, uint8_t* bytes, size_t cap, size_t* written, size_t* consumed,
                         size_t* bytes_written) {
  *written = *consumed = *bytes_written = 0;
  offsets[0] = 0;
  int32_t *prefixes = NULL, *suffixes = NULL;
  size_t n_prefix = 0, n_suffix = 0, used_prefix = 0, used_suffix = 0;
  int st = read_length_block(in, in_len, n, &prefixes, &n_prefix, &used_prefix);
  if (st != ORC_OK) return st;
  st = read_length_block(in + used_prefix, in_len - used_prefix, n, &suffixes, &n_suffix, &used_suffix);
  if (st != ORC_OK) {
    free(prefixes);
    return st;
  }
  if (cap > 0x7fffffffu) cap = 0x7fffffffu;
  size_t pos = used_prefix + used_suffix, out = 0, i = 0;
  const uint8_t* previous = prev;
  size_t previous_len = prev ? prev_len : 0;
  for (; i < n; ++i) {
    offsets[i] = (int32_t)out;
    if (i >= n_prefix || i >= n_suffix) {
      st = ORC_DECODE;
      break;
    }
    int32_t p = prefixes[i], s = suffixes[i];
    if (p < 0 || s < 0 || (size_t)s > in_len - pos || (size_t)p > previous_len) {
      st = ORC_DECODE;
      break;
    }
    if (out + (size_t)p + (size_t)s > cap) {
      st = ORC_BAD_ARG;
      break;
    }
    memmove(bytes + out, previous, (size_t)p);
    memcpy(bytes + out + (size_t)p, in + pos, (size_t)s);
    previous = bytes + out;
    previous_len = (size_t)p + (size_t)s;
    pos += (size_t)s;
    out += previous_len;
  }
  offsets[i] = (int32_t)out;
  free(prefixes);
  free(suffixes);
  *written = i;
  *consumed = pos;
  *bytes_written = out;
  return st;
}

/* ------------------------------------------------------------------------- */
/* Page decompression: SNAPPY and LZ4_RAW                                      */
/* ------------------------------------------------------------------------- */
/* The reference hands a page body to a codec of a third-party library and expects exactly
 * uncompressed_page_size bytes back (ColumnChunkPageReadStore.java:151-181,232-262, CodecFactory.java:
 * 176-195: BytesInput.from(stream, decompressedSize); any failure is "could not decompress page").
 * SNAPPY: SnappyDecompressor.java:27-35 calls org.xerial.snappy.Snappy.uncompress (snappy-java 1.1.10.8,
 * parquet-hadoop/pom.xml:145-147), the raw Snappy format: a varint32 with the uncompressed length, then
 * elements tagged by their low two bits - 00 literal (length-1 in the upper six bits, or 60..63 = that
 * many minus 59 little-endian length bytes), 01 copy of 4..11 bytes with an 11-bit offset, 10 copy of
 * 1..64 bytes with a 16-bit offset, 11 copy of 1..64 bytes with a 32-bit offset; copies may overlap
 * their own output.  LZ4_RAW: Lz4RawDecompressor.java:29-49 calls io.airlift.compress.lz4.Lz4Decompressor
 * (aircompressor 2.0.3, parquet-hadoop/pom.xml:157-161), the LZ4 block format: sequences of a token
 * (literal length in the high nibble, match length - 4 in the low one, 15 = add the following bytes up
 * to and including the first that is not 255), the literals, a 16-bit little-endian offset (0 is
 * invalid) and the match; the last sequence ends after its literals.
 * Neither library is in the reference tree: the restatement follows the published formats and is pinned
 * by pages compressed with the C libraries of the same formats (through pyarrow). */

#define ORC_CODEC_SNAPPY 1
#define ORC_CODEC_LZ4_RAW 2

static int lz_copy_match(uint8_t* out, size_t* op, size_t cap, size_t offset, size_t len) {
  if (offset == 0 || offset > *op || len > cap - *op) return ORC_DECODE;
  for (size_t i = 0; i < len; ++i) out[*op + i] = out[*op - offset + i]; /* byte by byte: overlaps repeat */
  *op += len;
  return ORC_OK;
}

int orc_snappy_decompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t cap, size_t* produced,
                          size_t* consumed) {
  size_t ip = 0, op = 0;
  *produced = *consumed = 0;
  /* preamble: varint32 */
  uint64_t ulen = 0;
  int shift = 0, ok = 0;
  while (ip < in_len && shift <= 28) {
    uint8_t b = in[ip++];
    ulen |= (uint64_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) {
      ok = 1;
      break;
    }
    shift += 7;
  }
  if (!ok || ulen > 0xFFFFFFFFull) return ORC_DECODE;
  int st = ORC_OK;
  while (ip < in_len) {
    uint8_t tag = in[ip];
    size_t hdr, len, offset = 0;
    switch (tag & 3) {
      case 0: {
        size_t l = tag >> 2;
        hdr = 1;
        if (l >= 60) {
          size_t nb = l - 59;
          if (in_len - ip < 1 + nb) {
            st = ORC_DECODE;
            goto done;
          }
          l = 0;
          for (size_t k = 0; k < nb; ++k) l |= (size_t)in[ip + 1 + k] << (8 * k);
          hdr = 1 + nb;
        }
        len = l + 1;
        if (len > in_len - ip - hdr || len > cap - op) {
          st = ORC_DECODE;
          goto done;
        }
        memcpy(out + op, in + ip + hdr, len);
        op += len;
        ip += hdr + len;
        continue;
      }
      case 1:
        hdr = 2;
        if (in_len - ip < hdr) {
          st = ORC_DECODE;
          goto done;
        }
        len = 4 + ((tag >> 2) & 7);
        offset = ((size_t)(tag >> 5) << 8) | in[ip + 1];
        break;
      case 2:
        hdr = 3;
        if (in_len - ip < hdr) {
          st = ORC_DECODE;
          goto done;
        }
        len = 1 + (tag >> 2);
        offset = (size_t)in[ip + 1] | ((size_t)in[ip + 2] << 8);
        break;
      default:
        hdr = 5;
        if (in_len - ip < hdr) {
          st = ORC_DECODE;
          goto done;
        }
        len = 1 + (tag >> 2);
        offset = (size_t)in[ip + 1] | ((size_t)in[ip + 2] << 8) | ((size_t)in[ip + 3] << 16) | ((size_t)in[ip + 4] << 24);
        break;
    }
    if ((st = lz_copy_match(out, &op, cap, offset, len))) goto done;
    ip += hdr;
  }
  /* the stream must produce what its preamble announced, and the page what its header announced */
  if (op != ulen || op != cap) st = ORC_DECODE;
done:
  *produced = op;
  *consumed = ip;
  return st;
}

int orc_lz4_raw_decompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t cap, size_t* produced,
                           size_t* consumed) {
  size_t ip = 0, op = 0;
  int st = ORC_OK;
  *produced = *consumed = 0;
  while (ip < in_len) {
    uint8_t token = in[ip++];
    size_t lit = token >> 4;
    if (lit == 15) {
      uint8_t b;
      do {
        if (ip >= in_len) {
          st = ORC_DECODE;
          goto done;
        }
        b = in[ip++];
        lit += b;
      } while (b == 255);
    }
    if (lit > in_len - ip || lit > cap - op) {
      st = ORC_DECODE;
      goto done;
    }
    memcpy(out + op, in + ip, lit);
    op += lit;
    ip += lit;
    if (ip == in_len) break; /* the last sequence has no match */
    if (in_len - ip < 2) {
      st = ORC_DECODE;
      goto done;
    }
    size_t offset = (size_t)in[ip] | ((size_t)in[ip + 1] << 8);
    ip += 2;
    size_t mlen = token & 15;
    if (mlen == 15) {
      uint8_t b;
      do {
        if (ip >= in_len) {
          st = ORC_DECODE;
          goto done;
        }
        b = in[ip++];
        mlen += b;
      } while (b == 255);
    }
    mlen += 4;
    if ((st = lz_copy_match(out, &op, cap, offset, mlen))) goto done;
  }
  if (op != cap) st = ORC_DECODE; /* the page header announced cap bytes */
done:
  *produced = op;
  *consumed = ip;
  return st;
}
