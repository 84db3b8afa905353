/*
 * pqd_oracle.h -- CPU restatement of parquet-java's column-page value decoders.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP kernels are compared
 * against; nothing in the shipped library links, loads or calls it.  Only the
 * tests, the smoke check and the benchmark's cpu_baseline leg may use it.
 *
 * Every function cites the reference file it restates (paths relative to the
 * apache/parquet-java tree).  The code is scalar and structured like the Java
 * (8 values per unpack step, per-run scratch buffers, per-value dictionary
 * lookup, eager whole-page delta / byte-stream-split decode) so that timing it
 * gives a fair "restated CPU" baseline.
 *
 * Pinning: the functions here are checked in tests/ against every known-answer
 * vector the reference's own tests hold for this path (bit strings of
 * TestBitPacking, header bytes and the decoder KAT of
 * TestRunLengthBitPackingHybridEncoder, float/double vectors of
 * ByteStreamSplitValuesReaderTest, ...), against the deterministic round-trip
 * patterns of the reference's tests, and against pages cut out of .parquet
 * files written by parquet-java itself.
 */
#ifndef PQD_ORACLE_H
#define PQD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes; numerically identical to the PQD_* codes of include/pqd.h. */
enum {
  ORC_OK = 0,
  ORC_EOF = 1,        /* java.io.EOFException: stream shorter than required          */
  ORC_BAD_ARG = 2,    /* IllegalArgumentException: bad width / count                 */
  ORC_PAST_END = 3,   /* "Reading past RLE/BitPacking stream."                       */
  ORC_DECODE = 4,     /* ParquetDecodingException / AIOOBE on malformed data         */
  ORC_DICT_OOB = 5    /* dictionary id >= dictionary size                            */
};

/* ---- bit packing (parquet-generator ByteBasedBitPackingGenerator.java:136-182,224-308) */
void orc_unpack8_32(int msb_first, int w, const uint8_t* in, int32_t* out);
void orc_unpack8_64(int msb_first, int w, const uint8_t* in, int64_t* out);
void orc_pack8_32(int msb_first, int w, const int32_t* in, uint8_t* out);
void orc_pack8_64(int msb_first, int w, const int64_t* in, uint8_t* out);

/* ParquetReadRouter.readBatch (parquet-plugins/.../ParquetReadRouter.java:107-116) and
 * readBatchUsing512Vector (:69-104): n (multiple of 8) w-bit LE values. */
int orc_read_batch_i32(int w, const uint8_t* in, size_t in_len, int32_t* out, size_t n,
                       size_t* consumed);
int orc_read_batch_i64(int w, const uint8_t* in, size_t in_len, int64_t* out, size_t n,
                       size_t* consumed);

/* ---- RLE / bit-packing hybrid (RunLengthBitPackingHybridDecoder.java:52-109) */
int orc_rle_decode(int w, const uint8_t* in, size_t in_len, int32_t* out, size_t n,
                   size_t* written, size_t* consumed);
/* RunLengthBitPackingHybridEncoder.java:146-273; returns encoded length or -1 */
long orc_rle_encode(int w, const int32_t* v, size_t n, uint8_t* out, size_t cap);
/* RunLengthBitPackingHybridValuesReader.java:41-45: 4-byte LE length prefix then the stream */
int orc_rle_prefixed_decode(int w, const uint8_t* in, size_t in_len, int32_t* out, size_t n,
                            size_t* written, size_t* consumed);

/* ---- dictionary pages (DictionaryValuesReader.java:49-127; writer DictionaryValuesWriter.java:153-180) */
int orc_dict_decode_ids(const uint8_t* page, size_t len, int32_t* ids, size_t n, size_t* written);
int orc_dict_decode(const uint8_t* page, size_t len, const void* dict, size_t dict_n,
                    int elem_size, void* out, size_t n, size_t* written);
long orc_dict_encode_ids(const int32_t* ids, size_t n, size_t dict_n, uint8_t* out, size_t cap);

/* ---- BINARY / FLBA / INT96 dictionaries (PlainValuesDictionary.java:88-115): entries as views
 * {low word: offset into the dictionary page, high word: length}; fixed_length 0 = length-prefixed */
#define ORC_VIEW_INVALID 0xffffffffffffffffULL /* an entry that only fails when it is read */
int orc_binary_dict_views(const uint8_t* page, size_t len, size_t n, int fixed_length, uint64_t* views);
/* readBytes() over n values whose views are known, gathered into offsets[n+1] + bytes */
int orc_binary_materialize(const uint64_t* views, size_t n, const uint8_t* dict_bytes, size_t dict_len,
                           int32_t* offsets, uint8_t* bytes, size_t cap, size_t* written,
                           size_t* bytes_written);

/* ---- DELTA_BINARY_PACKED (DeltaBinaryPackingValuesReader.java:60-172) */
int orc_delta_decode(const uint8_t* in, size_t in_len, int64_t* out, size_t cap,
                     size_t* total, size_t* consumed);
/* writers: DeltaBinaryPackingValuesWriterForLong.java:75-184 / ...ForInteger.java:75-182 */
long orc_delta_encode_i64(const int64_t* v, size_t n, int block, int mini, uint8_t* out, size_t cap);
long orc_delta_encode_i32(const int32_t* v, size_t n, int block, int mini, uint8_t* out, size_t cap);

/* ---- BYTE_STREAM_SPLIT (ByteStreamSplitValuesReader.java:53-218) */
int orc_bss_decode(const uint8_t* in, size_t in_len, int k, size_t value_count_bound,
                   uint8_t* out, size_t* n_values);
void orc_bss_encode(const uint8_t* in, size_t n, int k, uint8_t* out);

/* ---- legacy BIT_PACKED levels (ByteBitPackingValuesReader.java:41-88), MSB first */
int orc_bitpacked_levels_decode(int w, const uint8_t* in, size_t in_len, int32_t* out, size_t n,
                                size_t* consumed);

/* ---- PLAIN pages (pqd_oracle_pages.c): PlainValuesReader.java:38-145, BooleanPlainValuesReader.java:38-89 */
int orc_plain_decode(const uint8_t* in, size_t in_len, int elem, size_t n, uint8_t* out, size_t* written);
int orc_plain_boolean_decode(const uint8_t* in, size_t in_len, size_t n, uint8_t* out, size_t* written,
                             size_t* consumed);
/* ---- decoded levels -> validity / positions, null scatter, list offsets (ColumnReaderBase.java:642-668) */
void orc_def_validity(const int32_t* def, size_t n, int32_t max_def, uint8_t* validity, int32_t* value_index,
                      size_t* non_null);
int orc_scatter_nulls(const uint8_t* values, size_t n_values, const uint8_t* validity, size_t n_slots, int elem,
                      uint8_t* out, size_t* written, size_t* set_bits);
int orc_rep_offsets(const int32_t* rep, const int32_t* def, size_t n, int32_t slot_def, int32_t row_def, size_t rows_cap,
                    int32_t* offsets, uint8_t* row_valid, size_t* n_rows, size_t* n_slots);

/* ---- PLAIN BYTE_ARRAY (BinaryPlainValuesReader.java:36-74) -> offsets + bytes; bytes may be NULL */
int orc_plain_byte_array(const uint8_t* in, size_t in_len, size_t n, int32_t* offsets, uint8_t* bytes, size_t cap,
                         size_t* written, size_t* consumed, size_t* bytes_written);
/* ---- DELTA_LENGTH_BYTE_ARRAY (DeltaLengthByteArrayValuesReader.java:44-58) and DELTA_BYTE_ARRAY
 * (DeltaByteArrayReader.java:50-95) -> offsets + bytes; bytes may be NULL for the first (values stay in the page) */
int orc_delta_length_byte_array(const uint8_t* in, size_t in_len, size_t n, int32_t* offsets, uint8_t* bytes,
                                size_t cap, size_t* written, size_t* consumed, size_t* bytes_written);
int orc_delta_byte_array(const uint8_t* in, size_t in_len, size_t n, const uint8_t* prev, size_t prev_len,
                         int32_t* offsets, uint8_t* bytes, size_t cap, size_t* written, size_t* consumed,
                         size_t* bytes_written);

/* ---- page decompression (pqd_oracle_pages.c): raw Snappy and LZ4 block format; `cap` = the page header's
 * uncompressed size, which the stream must produce exactly */
int orc_snappy_decompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t cap, size_t* produced,
                          size_t* consumed);
int orc_lz4_raw_decompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t cap, size_t* produced,
                           size_t* consumed);
/* Zstandard frames (pqd_oracle_zstd.c, RFC 8878; no dictionaries) and the XXH64 of the content checksum */
int orc_zstd_decompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t cap, size_t* produced, size_t* consumed);
uint64_t orc_xxh64(const uint8_t* p, size_t n);
/* GZIP members (pqd_oracle_gzip.c, RFC 1952 / RFC 1951) and the CRC-32 of their trailers */
int orc_gzip_decompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t cap, size_t* produced, size_t* consumed);
uint32_t orc_crc32(uint32_t crc, const uint8_t* p, size_t n);

/* ---- BytesUtils helpers (parquet-common BytesUtils.java:202-269), exposed for tests */
int orc_read_uvarint32(const uint8_t* in, size_t len, size_t* pos, uint32_t* v);
int orc_read_zigzag_varlong(const uint8_t* in, size_t len, size_t* pos, int64_t* v);

#ifdef __cplusplus
}
#endif
#endif
