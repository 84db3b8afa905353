/*
 * pqd_oracle_pages.c -- CPU restatement of the steps on either side of the value decoders: PLAIN
 * pages, and what ColumnReaderBase does with decoded levels.
 *
 * TEST INFRASTRUCTURE ONLY (see pqd_oracle.h).  Each function cites the reference file it restates
 * (paths relative to the apache/parquet-java tree) and is written the way the Java is: one value
 * at a time.
 */
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
