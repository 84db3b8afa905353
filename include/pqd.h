/*
 * pqd.h -- C ABI of libpqdecode_hip.so: parquet column-page value decode on MI355X (gfx950).
 *
 * This is the drop-in boundary for parquet-java's page-decode path.  Every entry point is
 * `extern "C"`, takes plain pointers and sizes, and names the reference interface it replaces
 * (paths relative to the apache/parquet-java tree).  Java binds it through java.lang.foreign
 * (FFM) or JNI; Python tests bind it through ctypes.
 *
 * Conventions
 *   - Functions return a PQD_* status.  PQD_OK == 0.  No function aborts or throws.
 *   - "_dev" entry points take DEVICE pointers and enqueue work on `stream` (a hipStream_t passed
 *     as void*; NULL = the default stream) without synchronising.  Entry points without "_dev"
 *     take HOST pointers, stage the data through the device and return when the result is in the
 *     caller's buffer.
 *   - The library keeps no reference to caller memory after a host entry point returns, or after
 *     the stream work of a "_dev" entry point has completed.
 *   - Threads.  Every entry point may be called from any number of host threads at once
 *     (the reference's packers are shared singletons, parquet-encoding/.../Packer.java:113-121, and
 *     ParquetReadRouter's static methods are re-entrant, ParquetReadRouter.java:71-103; a
 *     ValuesReader instance itself stays single-threaded as in the reference).  The host-memory
 *     entry points (pqd_unpack_i32/_i64, pqd_decode_page, pqd_decode_page_flat,
 *     pqd_decode_binary_page) run on a stream and staging buffers owned by the calling thread;
 *     pqd_thread_release() gives them back before a thread ends.  "_dev" calls and
 *     pqd_decode_pages issued on the SAME stream from several threads are serialised inside the
 *     library for as long as their launches are being enqueued; calls on different streams run
 *     concurrently.
 *   - All bit-packed data is parquet's little-endian (LSB-first) packing unless stated otherwise.
 */
#ifndef PQD_H
#define PQD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PQD_VERSION 100 /* 0.1.0 */

/* ---- status codes and the Java exception each one stands for ------------------------------ */
enum pqd_status {
  PQD_OK = 0,
  PQD_EOF = 1,       /* java.io.EOFException: fewer bytes than the decode needs                  */
  PQD_BAD_ARG = 2,   /* IllegalArgumentException: bit width / count outside the contract          */
  PQD_PAST_END = 3,  /* IllegalArgumentException("Reading past RLE/BitPacking stream.")           */
  PQD_DECODE = 4,    /* ParquetDecodingException: malformed page                                  */
  PQD_DICT_OOB = 5,  /* ArrayIndexOutOfBounds -> ParquetDecodingException: id >= dictionary size  */
  PQD_HIP_ERROR = 6, /* a HIP runtime call failed; see pqd_last_error()                           */
  PQD_NO_DEVICE = 7  /* no usable gfx950 device                                                  */
};

/* ---- page encodings (org.apache.parquet.column.Encoding, parquet-column/.../Encoding.java:62-304) */
enum pqd_encoding {
  PQD_ENC_RLE = 1,                 /* RLE / bit-packing hybrid -> int32 (levels, booleans, dictionary ids) */
  PQD_ENC_RLE_DICTIONARY = 2,      /* 1-byte width + hybrid ids, gathered through a dictionary (also PLAIN_DICTIONARY data pages) */
  PQD_ENC_DELTA_BINARY_PACKED = 3, /* -> int32 / int64.  Limits of this library: one miniblock's packed bytes
                                      (miniblock size x width / 8) must fit 18 KiB and a block may hold at
                                      most 256 miniblocks, else PQD_DECODE; parquet-java's writers use 4
                                      miniblocks of 32 values: 256 bytes at most */
  PQD_ENC_BYTE_STREAM_SPLIT = 4,   /* -> elem_size-byte elements */
  PQD_ENC_BIT_PACKED = 5,          /* legacy MSB-first levels -> int32 */
  PQD_ENC_PLAIN = 6,               /* fixed-width PLAIN values: elem_size-byte elements, 1..16 (INT32 / INT64 / FLOAT /
                                      DOUBLE / INT96 / FIXED_LEN_BYTE_ARRAY), copied as they are:
                                      PlainValuesReader.java:38-145, FixedLenByteArrayPlainValuesReader.java.
                                      A page with fewer than num_values values: PQD_DECODE behind the last whole one */
  PQD_ENC_PLAIN_BOOLEAN = 7,       /* PLAIN booleans, bit (i & 7) of byte (i >> 3): BooleanPlainValuesReader.java:38-89
                                      -> one byte (0 / 1) per value (elem_size 1) or one int32 (elem_size 4);
                                      bytes_consumed = min(ceil(num_values / 8), data_len) */
  PQD_ENC_DELTA_LENGTH_BYTE_ARRAY = 8, /* pqd_delta_byte_array_dev only */
  PQD_ENC_DELTA_BYTE_ARRAY = 9,        /* pqd_delta_byte_array_dev only */
  PQD_ENC_PLAIN_BYTE_ARRAY = 10        /* pqd_delta_byte_array_dev only: [length: 4 bytes LE][bytes] ... */
};

/* pqd_page_desc.flags */
#define PQD_F_LENGTH_PREFIX 1u /* PQD_ENC_RLE: 4-byte LE length in front (V1 levels, RLE booleans):
                                  RunLengthBitPackingHybridValuesReader.java:41-45 */
#define PQD_F_WIDTH_PREFIX 2u  /* PQD_ENC_RLE: 1-byte bit width in front (dictionary ids as ints):
                                  DictionaryValuesReader.java:53, readValueDictionaryId :67-73 */

/*
 * One page (or level stream) to decode.  Replaces ValuesReader.initFromPage(valueCount, stream)
 * followed by num_values calls of readInteger/readLong/readFloat/readDouble/readValueDictionaryId
 * (parquet-column/.../values/ValuesReader.java:107-202).  The same struct is used with host
 * pointers (pqd_decode_page) and device pointers (pqd_decode_pages_dev).
 */
typedef struct pqd_page_desc {
  const uint8_t* data; /* first byte of the stream                                               */
  void* out;           /* num_values elements of elem_size bytes                                 */
  const void* dict;    /* PQD_ENC_RLE_DICTIONARY: dict_len values of elem_size bytes (PLAIN
                          dictionary page body, PlainValuesDictionary.java:148-157,232-241)      */
  uint32_t data_len;   /* bytes available from `data` (ByteBufferInputStream.available())        */
  uint32_t num_values; /* values to produce = capacity of `out` in elements.  For the hybrid
                          encodings: the number of read*() calls; for DELTA / BYTE_STREAM_SPLIT:
                          the page's valueCount upper bound                                      */
  uint32_t dict_len;   /* dictionary entries                                                     */
  uint8_t encoding;    /* enum pqd_encoding                                                      */
  uint8_t elem_size;   /* bytes per output element: 4 or 8 (BYTE_STREAM_SPLIT: 1..16)            */
  uint8_t bit_width;   /* PQD_ENC_RLE without WIDTH_PREFIX, PQD_ENC_BIT_PACKED                   */
  uint8_t flags;       /* PQD_F_*                                                                */
} pqd_page_desc;

typedef struct pqd_page_result {
  int32_t status;          /* PQD_* of the first failure in value order, or PQD_OK              */
  uint32_t values_written; /* leading elements of `out` that are valid                          */
  uint32_t bytes_consumed; /* stream bytes used (ValuesReader.getNextOffset semantics where the
                              reference defines them: DELTA_BINARY_PACKED, length-prefixed RLE,
                              BIT_PACKED, BYTE_STREAM_SPLIT)                                     */
  uint32_t reserved;
} pqd_page_result;

/* ---- runtime ------------------------------------------------------------------------------ */
int pqd_version(void);
const char* pqd_status_string(int status);
/* Text of the last failure on the calling thread ("" if none). */
const char* pqd_last_error(void);
/* Number of gfx950 devices visible (0 if none / no driver). */
int pqd_device_count(void);
/* Makes `device` current for the calling thread and initialises per-device state.
 * Mirrors the reference's "plugin present?" probe (parquet-encoding/.../Packer.java:68-80,
 * ParquetReadRouter.java:50-54): PQD_NO_DEVICE plays the role of VectorSupport.NONE. */
int pqd_init(int device);

int pqd_dev_alloc(void** p, size_t bytes);
int pqd_dev_free(void* p);
int pqd_host_alloc_pinned(void** p, size_t bytes);
int pqd_host_free_pinned(void* p);
/* the same two under their shorter names */
int pqd_alloc_pinned(void** p, size_t bytes);
int pqd_free_pinned(void* p);
int pqd_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);
int pqd_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);
int pqd_memset_dev(void* dst_dev, int value, size_t bytes, void* stream);
int pqd_stream_create(void** stream);
int pqd_stream_destroy(void* stream);
int pqd_stream_sync(void* stream);
/* HIP events on the library's streams (for timing the kernels where they run). */
int pqd_event_create(void** ev);
int pqd_event_destroy(void* ev);
int pqd_event_record(void* ev, void* stream);
int pqd_event_sync(void* ev);
int pqd_event_elapsed_ms(void* start, void* stop, float* ms);
/*
 * Per-kernel timing of pqd_decode_pages_dev for dictionary pages, which is a sequence of up to
 * three launches (run-plan kernel, sliced gather kernel, generic kernel for the pages left over).
 * While enabled, the calling thread's calls (at most 4096 of them) record HIP events between the
 * launches on the call's stream.  pqd_phase_ms_mean waits for those calls, returns the mean
 * duration of each of the three launches in milliseconds and the number of calls averaged, and
 * starts a new series (disabling only pauses the recording).  enable = 2 times the gather launch alone,
 * with two events that the launch itself records at the kernel's start and end (ms[0] and ms[2] then
 * read 0): an event recorded between two launches costs the stream a few microseconds.
 * pqd_decode_binary_pages_dev is timed the same way (its one launch is ms[1]).  pqd_set_phase_series selects one
 * of four independent series (0..3, 0 at first) that the calls after it record into and pqd_phase_ms_mean reads:
 * a caller that alternates calls of different kinds keeps a series per kind.  No counterpart in the reference;
 * off by default.
 */
int pqd_set_phase_timing(int enable);
int pqd_set_phase_series(int series);
int pqd_phase_ms_mean(float ms[3], int* n_calls);

/* ---- bulk bit-unpack: one bit-packed run --------------------------------------------------- */
/*
 * Replaces ParquetReadRouter.read / readBatchUsing512Vector / readBatch(bitWidth, in,
 * currentCount, currentBuffer)  (parquet-plugins/parquet-encoding-vector/.../ParquetReadRouter.java:57-116)
 * and, per step, BytePacker.unpack8Values / unpack32Values / unpackValuesUsingVector
 * (parquet-encoding/.../BytePacker.java:87-148).
 * n_values must be a multiple of 8; bit_width 0..32; exactly n_values*bit_width/8 bytes are
 * consumed; PQD_EOF if in_len is smaller (the Java in.slice() throws EOFException).
 * Width 32 returns raw bit patterns (negative ints are legal).
 */
int pqd_unpack_i32(int bit_width, const uint8_t* in, size_t in_len, int32_t* out, size_t n_values);
int pqd_unpack_i32_dev(int bit_width, const uint8_t* in, size_t in_len, int32_t* out, size_t n_values,
                       void* stream);
/* BytePackerForLong.unpack8Values / unpack32Values over a whole run, widths 0..64
 * (parquet-encoding/.../BytePackerForLong.java:72-105). */
int pqd_unpack_i64(int bit_width, const uint8_t* in, size_t in_len, int64_t* out, size_t n_values);
int pqd_unpack_i64_dev(int bit_width, const uint8_t* in, size_t in_len, int64_t* out, size_t n_values,
                       void* stream);

/* ---- page decode ---------------------------------------------------------------------------- */
/*
 * Device-resident batch: `descs` and `results` are DEVICE arrays of n_pages entries whose
 * data/out/dict members are device pointers.  One workgroup decodes one page.  All pages of one
 * call must share `encoding` and `elem_size` (one launch per encoding class); a descriptor that
 * disagrees gets PQD_BAD_ARG in its result.  Replaces, per page, Encoding.getValuesReader /
 * getDictionaryBasedValuesReader + initFromPage + read loop as driven by
 * ColumnReaderBase.initDataReader/readPageV1/readPageV2 (parquet-column/.../impl/ColumnReaderBase.java:693-780).
 */
int pqd_decode_pages_dev(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages, int encoding,
                         int elem_size, void* stream);
/*
 * pqd_decode_pages_dev for callers that cannot allocate device memory while they run: code under
 * HIP graph capture, serving loops that must not touch the allocator, harnesses that forbid it.
 * (No counterpart in the reference: the JVM allocates per run, RunLengthBitPackingHybridDecoder.java:94-99.)
 * Same arguments, same results, same kernels; the scratch of the planned hybrid / dictionary paths
 * (chunk pool 16 MiB, plan records 16 KiB, counters and slot table 6.25 KiB, eight dictionary plane slots
 * 4 MiB: 20.02 MiB in all) lives in a device array that is part of
 * the library image, one copy per GPU, instead of buffers the library allocates.  It calls no
 * hipMalloc / hipFree.  Limits: at most 2048 pages per call (PQD_BAD_ARG beyond that; pages whose
 * runs outgrow the 8192-chunk pool are decoded by the generic kernel, as in pqd_decode_pages_dev).
 * All calls on one GPU share that GPU's array: a call may be enqueued only when the previous
 * call's work on that GPU has finished or is ahead of it on the same stream; calls that overlap on
 * one GPU (different streams) race.  Encodings that use no scratch (DELTA_BINARY_PACKED,
 * BYTE_STREAM_SPLIT, BIT_PACKED) are passed through to pqd_decode_pages_dev.
 */
int pqd_decode_pages_dev_noalloc(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages, int encoding,
                                 int elem_size, void* stream);
/*
 * Host-side batch: `descs`/`results` are HOST arrays, but data/out/dict are still device pointers.
 * Groups the pages by encoding class, uploads the descriptors, launches, downloads the results and
 * synchronises `stream`.
 */
int pqd_decode_pages(const pqd_page_desc* descs, pqd_page_result* results, uint32_t n_pages, void* stream);
/*
 * One page, everything in host memory (what a Java ValuesReader subclass calls from
 * initFromPage): stages data (+ dictionary) to the device, decodes, copies `out` back.
 */
int pqd_decode_page(const pqd_page_desc* desc, pqd_page_result* result);
/*
 * The same call with every pointer as a direct argument instead of a struct member, for bindings
 * that can only pass heap arrays as arguments (java.lang.foreign Linker.Option.critical(true) with
 * MemorySegment.ofArray, JNI GetPrimitiveArrayCritical): `data`, `dict` and `out` may then be the
 * Java byte[] / int[] / long[] themselves (ParquetReadRouter.java:80-103 takes heap and direct
 * buffers alike).  Arguments have the meaning of the pqd_page_desc members of the same name.
 */
int pqd_decode_page_flat(int encoding, int elem_size, int bit_width, uint32_t flags, const uint8_t* data,
                         uint32_t data_len, uint32_t num_values, const void* dict, uint32_t dict_len, void* out,
                         pqd_page_result* result);
/*
 * Frees what the host-memory entry points keep for the calling thread (its stream, device staging
 * buffers, plan scratch, timing events).  Optional: a thread that ends without it releases the same
 * things when the C++ runtime runs its thread-local destructors; call it from runtimes that end
 * threads without running those (or before unloading the library).  The thread may go on calling
 * the library afterwards.  No counterpart in the reference (the JVM collects its buffers).
 */
int pqd_thread_release(void);

/* ---- dictionaries ----------------------------------------------------------------------------- */
/* org.apache.parquet.format.Type numbering (PrimitiveTypeName of the column). */
enum pqd_physical_type {
  PQD_TYPE_INT32 = 1,
  PQD_TYPE_INT64 = 2,
  PQD_TYPE_INT96 = 3,
  PQD_TYPE_FLOAT = 4,
  PQD_TYPE_DOUBLE = 5,
  PQD_TYPE_BYTE_ARRAY = 6,
  PQD_TYPE_FIXED_LEN_BYTE_ARRAY = 7
};

/* A view of one BINARY value, as Binary.fromConstantByteBuffer(dictionaryBytes, offset, length)
 * holds it: low 32 bits = offset of its first byte in the dictionary page, high 32 bits = length. */
#define PQD_VIEW_INVALID 0xffffffffffffffffULL /* an entry that fails when it is read (see below) */

/*
 * HOST: walks a PLAIN dictionary page of BYTE_ARRAY entries ([length: 4 bytes LE][bytes] ..., or
 * fixed_length bytes each when fixed_length > 0: FIXED_LEN_BYTE_ARRAY, INT96 = 12) and writes one
 * view per entry.  Replaces the constructor PlainBinaryDictionary(dictionaryPage, length)
 * (parquet-column/.../values/dictionary/PlainValuesDictionary.java:88-115).  Like the reference it
 * reads every length prefix (one outside the page -> PQD_DECODE, Java: IndexOutOfBoundsException)
 * but does not look at the value bytes: an entry that runs past the page or has a negative length
 * becomes PQD_VIEW_INVALID and fails only when a page refers to it.
 */
int pqd_binary_dict_views(const uint8_t* page, size_t page_len, uint32_t n_entries, int fixed_length,
                          uint64_t* views);

/*
 * Dictionary handle: the decoded dictionary page, resident on the current device.  Replaces
 * Encoding.PLAIN.initDictionary(descriptor, dictionaryPage) (parquet-column/.../Encoding.java:89-112)
 * -> PlainValuesDictionary.Plain{Integer,Long,Float,Double,Binary}Dictionary
 * (PlainValuesDictionary.java:59-345).  `page` is the HOST dictionary page body, n_entries its
 * dictionarySize.  INT32/FLOAT/INT64/DOUBLE: entries are the 4-/8-byte values (a page shorter than
 * n_entries values -> PQD_DECODE).  BYTE_ARRAY / FIXED_LEN_BYTE_ARRAY (type_length) / INT96:
 * entries are 8-byte views into `bytes`, the page body.  Synchronous.
 */
typedef struct pqd_dict pqd_dict;
typedef struct pqd_dict_info {
  const void* entries;  /* DEVICE: n_entries elements of elem_size bytes -> pqd_page_desc.dict        */
  const uint8_t* bytes; /* DEVICE: the dictionary page body (BINARY kinds), else NULL                 */
  uint32_t n_entries;   /* -> pqd_page_desc.dict_len; Dictionary.getMaxId() + 1                       */
  uint32_t elem_size;   /* 4 or 8 -> pqd_page_desc.elem_size                                          */
  uint32_t bytes_len;
  uint32_t physical_type;
} pqd_dict_info;
int pqd_dict_create(int physical_type, int type_length, const uint8_t* page, size_t page_len, uint32_t n_entries,
                    pqd_dict** dict);
int pqd_dict_get(const pqd_dict* dict, pqd_dict_info* info);
int pqd_dict_destroy(pqd_dict* dict);

/*
 * BINARY values of a page, gathered: replaces the loop `Binary b = reader.readBytes(); copy b`
 * (DictionaryValuesReader.readBytes, parquet-column/.../values/dictionary/DictionaryValuesReader.java:76-83,
 * Dictionary.decodeToBinary) for a page whose values are already known as views (a
 * PQD_ENC_RLE_DICTIONARY decode with elem_size 8 and the dictionary's views as `dict`, which is the
 * reference's own zero-copy result).  offsets[i] = start of value i in `bytes`,
 * offsets[values_written] = end of the last value delivered (= result.bytes_consumed).
 * PQD_DECODE at the first value whose view is PQD_VIEW_INVALID or outside dict_bytes; PQD_BAD_ARG at
 * the first value that does not fit into bytes_capacity (at most 2^31-1 bytes per page: the offsets
 * are Java ints).  All pointers are DEVICE pointers.
 */
typedef struct pqd_binary_page_desc {
  const uint64_t* views;     /* num_values views                                                   */
  const uint8_t* dict_bytes; /* the dictionary page body the views point into                      */
  int32_t* offsets;          /* num_values + 1                                                     */
  uint8_t* bytes;            /* bytes_capacity                                                     */
  uint32_t num_values;
  uint32_t dict_bytes_len;
  uint32_t bytes_capacity;
  uint32_t reserved;
} pqd_binary_page_desc;
int pqd_binary_materialize_dev(const pqd_binary_page_desc* descs, pqd_page_result* results, uint32_t n_pages,
                               void* stream);
/*
 * The two steps above in one: dictionary data pages of a BINARY column (BYTE_ARRAY, FIXED_LEN_BYTE_ARRAY,
 * INT96) -> offsets + bytes, without the views in between.  Replaces the read loop over
 * DictionaryValuesReader.readBytes (DictionaryValuesReader.java:76-83: id = decoder.readInt();
 * dictionary.decodeToBinary(id), PlainValuesDictionary.java:88-118) for whole pages.
 * pages[i]: the data page as for pqd_decode_pages_dev (encoding PQD_ENC_RLE_DICTIONARY, elem_size 8, dict_len;
 * `dict` and `out` are not used); bins[i]: where the page's values go (offsets, bytes, bytes_capacity) and
 * dict_bytes / dict_bytes_len = pqd_dict_info.bytes / bytes_len of `dict` (views and num_values are not used).
 * page_results[i] is what pqd_decode_pages_dev reports for the page's ids (status, values the read loop
 * gets before it fails: PQD_PAST_END, PQD_DICT_OOB, ...); bin_results[i] what pqd_binary_materialize_dev reports
 * for those values (values_written, bytes_consumed = bytes written; PQD_DECODE for a broken dictionary entry,
 * PQD_BAD_ARG at the first value that does not fit).  Differences from the two steps: when a page stops at a
 * broken entry or for lack of room, the id stream is not read further and page_results[i] is {PQD_OK, the
 * values delivered, the stream bytes read so far}; after PQD_DICT_OOB bytes_consumed likewise counts the
 * bytes read so far.  All pointers but `dict` are DEVICE pointers; asynchronous on `stream`.
 */
int pqd_decode_binary_pages_dev(const pqd_dict* dict, const pqd_page_desc* pages, const pqd_binary_page_desc* bins,
                                pqd_page_result* page_results, pqd_page_result* bin_results, uint32_t n_pages,
                                void* stream);
/*
 * One BINARY dictionary data page, everything but the dictionary in HOST memory (what a Java
 * DictionaryValuesReader over a PlainBinaryDictionary calls from initFromPage): decodes the ids,
 * gathers the views and the bytes, copies offsets[0..values_written] and the bytes back.  The
 * status is that of the first value that fails, in value order (PQD_PAST_END, PQD_DICT_OOB,
 * PQD_DECODE, PQD_BAD_ARG); the values before it are delivered.
 */
int pqd_decode_binary_page(const pqd_dict* dict, const uint8_t* page, uint32_t page_len, uint32_t num_values,
                           int32_t* offsets, uint8_t* bytes, uint32_t bytes_capacity, pqd_page_result* result);

/* ---- PLAIN BYTE_ARRAY / DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY value pages ------------------------- */
/*
 * Replaces the read loops over BinaryPlainValuesReader (parquet-column/.../values/plain/
 * BinaryPlainValuesReader.java:36-74: every value is its length in four little-endian bytes and then its bytes;
 * a value that cannot be read fails with "could not read bytes at offset"), DeltaLengthByteArrayValuesReader (parquet-column/.../values/
 * deltalengthbytearray/DeltaLengthByteArrayValuesReader.java:33-77: all lengths as one
 * DELTA_BINARY_PACKED block, then the values back to back; readBytes slices `length` bytes and fails
 * with "Failed to read N bytes" when the page ends first) and DeltaByteArrayReader (.../deltastrings/
 * DeltaByteArrayReader.java:32-95: prefix lengths as one DELTA_BINARY_PACKED block, then a
 * DELTA_LENGTH_BYTE_ARRAY section of suffixes; value i = first prefix[i] bytes of value i-1 + suffix i.
 * The prefix of a page's first value refers to the last value of the page before it when the caller
 * links them, as ColumnReaderBase does through setPreviousReader for files written before PARQUET-246
 * was fixed; current writers start every page with prefix 0).
 * Output as for pqd_binary_materialize_dev: offsets[i] = start of value i in `bytes`,
 * offsets[values_written] = end of the last value delivered = result.reserved; result.bytes_consumed =
 * page bytes read (length blocks + value bytes).  With bytes == NULL a DELTA_LENGTH_BYTE_ARRAY page is
 * left where it is (the reference's zero-copy Binary views): value i is
 * data[bytes_consumed - reserved + offsets[i] ...]; a PLAIN page likewise: value i is
 * data[offsets[i] + 4 * (i + 1) ...].  PLAIN pages: PQD_DECODE at the first value whose length field or bytes do not
 * lie inside the page or whose length is negative; bytes_consumed = the page bytes in front of the first value
 * not delivered; `work` is not used (may be NULL); pages above 2^31-1 bytes are PQD_BAD_ARG.
 * Status: a length block that fails fails the page before its first value (both readers decode their
 * length blocks in initFromPage); PQD_DECODE at the first value whose length is negative, whose bytes
 * run past the page, whose prefix is longer than the value before it, or that has no length left in
 * a block ("no more value to read"); PQD_BAD_ARG at the first value that does not fit into
 * bytes_capacity (at most 2^31-1).  (A negative length has no defined meaning in the reference: an
 * empty value from MultiBufferInputStream.slice, a backwards seek from SingleBufferInputStream.slice.)
 * All pointers are DEVICE pointers.  `work` is scratch of PQD_DELTA_BYTES_WORK(num_values) bytes per
 * page, 16-byte aligned, that the call overwrites.
 */
#define PQD_DELTA_BYTES_WORK(num_values) (64u + 8u * (size_t)(num_values))
typedef struct pqd_delta_bytes_desc {
  const uint8_t* data;         /* the page's values section                                          */
  int32_t* offsets;            /* num_values + 1                                                     */
  uint8_t* bytes;              /* bytes_capacity                                                     */
  uint8_t* work;               /* PQD_DELTA_BYTES_WORK(num_values) bytes                             */
  const int32_t* prev_offsets; /* PQD_ENC_DELTA_BYTE_ARRAY: offsets / bytes of the page decoded      */
  const uint8_t* prev_bytes;   /* before this one (an earlier call on the same stream); value        */
                               /* prev_count-1 is the value the first prefix refers to.  NULL: none  */
  uint32_t data_len;
  uint32_t num_values;
  uint32_t bytes_capacity;
  uint32_t encoding;           /* PQD_ENC_PLAIN_BYTE_ARRAY, _DELTA_LENGTH_BYTE_ARRAY or _DELTA_BYTE_ARRAY */
  uint32_t prev_count;
  uint32_t reserved;
} pqd_delta_bytes_desc;
int pqd_delta_byte_array_dev(const pqd_delta_bytes_desc* descs, pqd_page_result* results, uint32_t n_pages, void* stream);
/*
 * One such page in HOST memory (what initFromPage + the read loop of the two readers see); `previous`
 * is the last value of the page before (DELTA_BYTE_ARRAY only, may be NULL / 0).  *bytes_needed (may
 * be NULL) receives the sum of all lengths in the page's length blocks (PLAIN: of the values that can be
 * read): the capacity with which a call that ended in PQD_BAD_ARG succeeds.  Thread-safe like the other host entry points.
 */
int pqd_decode_delta_byte_array_page(int encoding, const uint8_t* page, uint32_t page_len, uint32_t num_values,
                                     const uint8_t* previous, uint32_t previous_len, int32_t* offsets, uint8_t* bytes,
                                     uint32_t bytes_capacity, pqd_page_result* result, uint64_t* bytes_needed);

/* ---- levels -> columnar structure ---------------------------------------------------------------- */
/*
 * The reference assembles records value by value: ColumnReaderBase.checkRead takes one repetition and
 * one definition level per value; the value is present exactly when dl == maxDefinitionLevel and a
 * record starts where rl == 0 (parquet-column/.../impl/ColumnReaderBase.java:642-668, isNull :322-325).
 * The three entry points below give a columnar consumer the same facts as arrays, from levels that
 * pqd_decode_pages_dev (PQD_ENC_RLE / PQD_ENC_BIT_PACKED) has decoded to int32.  All pointers are
 * DEVICE pointers; descriptor and result arrays are device arrays of n_pages entries; one page per
 * descriptor.  pqd_page_result.reserved carries the count named below.
 */
/* Definition levels -> validity.  validity bit i (LSB first, Arrow's layout; ceil(n_levels / 8) bytes,
 * padding bits zero) = def_levels[i] == max_def; value_index[i] = position of slot i's value among the
 * page's non-null values, -1 for a null.  Either output may be NULL.  def_levels NULL = a required
 * column (max level 0: Encoding.java:122-125, every value present).
 * result: values_written = n_levels, reserved = number of non-null values (= the values the page's
 * data section holds: the num_values to decode it with). */
typedef struct pqd_validity_desc {
  const int32_t* def_levels;
  uint8_t* validity;
  int32_t* value_index;
  uint32_t n_levels;
  int32_t max_def;
} pqd_validity_desc;
int pqd_def_levels_to_validity_dev(const pqd_validity_desc* descs, pqd_page_result* results, uint32_t n_pages, void* stream);
/* Null scatter: out[i] = validity bit i ? values[rank of i among the set bits] : zero bytes, for n_slots
 * slots of elem_size bytes (1..16; all pages of a call share it).  validity NULL = all set.
 * result: values_written = n_slots, reserved = set bits; a set bit whose value lies beyond n_values:
 * PQD_DECODE, values_written = that slot (slots before it are right, later ones hold zeros where the
 * values are missing). */
typedef struct pqd_scatter_desc {
  const void* values;      /* the page's dense non-null values, n_values of them */
  const uint8_t* validity; /* n_slots bits */
  void* out;               /* n_slots elements */
  uint32_t n_slots;
  uint32_t n_values;
} pqd_scatter_desc;
int pqd_scatter_nulls_dev(const pqd_scatter_desc* descs, pqd_page_result* results, uint32_t n_pages, int elem_size,
                          void* stream);
/* Repetition levels -> list offsets, for the outermost repeated level: row r starts at the r-th level
 * with rl == 0; a level holds an element slot (null or not) when dl >= slot_def; offsets[r] = element
 * slots before row r, offsets[n_rows] = all of them.  row_valid[r] (optional, one byte per row) = the
 * definition level at the row's first level >= row_def (the list itself is not null).
 * result: values_written = n_rows, reserved = element slots.  More rows than rows_capacity:
 * PQD_BAD_ARG, the first rows_capacity offsets are written. */
typedef struct pqd_offsets_desc {
  const int32_t* rep_levels;
  const int32_t* def_levels;
  int32_t* offsets;       /* rows_capacity + 1 */
  uint8_t* row_valid;     /* rows_capacity, or NULL */
  uint32_t n_levels;
  uint32_t rows_capacity;
  int32_t slot_def;
  int32_t row_def;
} pqd_offsets_desc;
int pqd_rep_levels_to_offsets_dev(const pqd_offsets_desc* descs, pqd_page_result* results, uint32_t n_pages, void* stream);

/* ---- column chunks: page headers -> page table -> descriptors ------------------------------------- */
/*
 * HOST.  Walks the pages of one column chunk: replaces ParquetFileReader.Chunk.readAllPages
 * (parquet-hadoop/.../hadoop/ParquetFileReader.java:1973-2117) up to, but not including, decompression.
 * `chunk` = the chunk's bytes from its first page header (dictionary page first, if there is one);
 * value_count = ColumnChunkMetaData.getValueCount(): pages are read while fewer values have been seen
 * (hasMorePages, :2130-2134); -1 = until the bytes are used up.  Each page's thrift compact PageHeader
 * (parquet-format PageHeader / DataPageHeader / DictionaryPageHeader / DataPageHeaderV2) is parsed into
 * one pqd_chunk_page; pages of other types (INDEX_PAGE) are skipped as the reference skips them.
 * verify_crc != 0: a page whose header carries a crc is checked (CRC-32 of the page bytes as stored;
 * V2: repetition levels + definition levels + values), as with usePageChecksumVerification (:2021-2046).
 * Errors (Java: IOException / ParquetDecodingException): PQD_EOF header or page runs past the chunk;
 * PQD_DECODE malformed header, a second dictionary page ("more than one dictionary page in column"),
 * CRC mismatch, negative sizes, level lengths beyond the page, or a value count that does not add up
 * ("Expected N values in column chunk ... but got M").  PQD_BAD_ARG: more pages than pages_capacity
 * (n_pages then holds the number needed).
 */
enum pqd_page_type { PQD_PAGE_DATA = 0, PQD_PAGE_INDEX = 1, PQD_PAGE_DICTIONARY = 2, PQD_PAGE_DATA_V2 = 3 };
/* org.apache.parquet.format.Encoding numbering, as page headers carry it */
enum pqd_format_encoding {
  PQD_FMT_PLAIN = 0, PQD_FMT_PLAIN_DICTIONARY = 2, PQD_FMT_RLE = 3, PQD_FMT_BIT_PACKED = 4,
  PQD_FMT_DELTA_BINARY_PACKED = 5, PQD_FMT_DELTA_LENGTH_BYTE_ARRAY = 6, PQD_FMT_DELTA_BYTE_ARRAY = 7,
  PQD_FMT_RLE_DICTIONARY = 8, PQD_FMT_BYTE_STREAM_SPLIT = 9
};
typedef struct pqd_chunk_page {
  uint32_t header_offset;     /* of the page header in the chunk                                        */
  uint32_t body_offset;       /* of the page bytes behind it                                            */
  uint32_t compressed_size;   /* bytes stored (PageHeader.compressed_page_size)                         */
  uint32_t uncompressed_size; /* PageHeader.uncompressed_page_size                                      */
  uint32_t num_values;        /* incl. nulls (data pages); dictionary entries (dictionary page)         */
  uint32_t num_nulls;         /* V2                                                                     */
  uint32_t num_rows;          /* V2                                                                     */
  uint32_t rep_levels_len;    /* V2: repetition_levels_byte_length                                      */
  uint32_t def_levels_len;    /* V2: definition_levels_byte_length                                      */
  uint32_t crc;               /* valid when has_crc                                                     */
  uint8_t type;               /* enum pqd_page_type                                                     */
  uint8_t encoding;           /* enum pqd_format_encoding of the values                                 */
  uint8_t def_encoding;       /* V1 (V2: RLE)                                                           */
  uint8_t rep_encoding;       /* V1 (V2: RLE)                                                           */
  uint8_t is_compressed;      /* V2: DataPageHeaderV2.is_compressed (default true); V1 / dictionary: 1  */
  uint8_t has_crc;
  uint8_t reserved[2];
} pqd_chunk_page;
int pqd_chunk_walk(const uint8_t* chunk, size_t chunk_len, int64_t value_count, int verify_crc, pqd_chunk_page* pages,
                   uint32_t pages_capacity, uint32_t* n_pages);
/* CRC-32 (java.util.zip.CRC32, the polynomial of zlib) of `len` bytes, continuing from `crc` (0 to start). HOST. */
uint32_t pqd_crc32(uint32_t crc, const uint8_t* data, size_t len);

/*
 * What ColumnReaderBase.readPageV1 / readPageV2 / initDataReader do with a page once it is uncompressed
 * (parquet-column/.../impl/ColumnReaderBase.java:693-780), for all data pages of a chunk at once and
 * on the DEVICE: each page's body is cut into repetition levels, definition levels and values and the
 * three decode descriptors are written.  V1: the level sections carry their own 4-byte lengths (RLE,
 * RunLengthBitPackingHybridValuesReader.java:41-45) or are ceil(num_values * width / 8) bytes
 * (BIT_PACKED, ByteBitPackingValuesReader.java:78-85), read from the body on the device; V2: lengths
 * from the header, no prefix (:751-780).  A level of maximum 0 has no bytes (Encoding.java:122-125).
 *   sections[i]   page table entry i (DEVICE copy of the data pages of pqd_chunk_walk, dictionary page
 *                 left out) with body_offset = offset of the page's UNCOMPRESSED body in `bodies`
 *   rep / def     descriptors of the level streams -> int32 levels at rep_out / def_out + first value
 *                 index of the page (level_base[i], in values); encoding PQD_ENC_RLE or PQD_ENC_BIT_PACKED,
 *                 data_len 0 for a level of maximum 0 (such a page's levels are all zero: the caller
 *                 clears the outputs once)
 *   val           descriptor of the values: encoding and element size from the page's encoding and the
 *                 column's physical type; num_values = the page's value count incl. nulls (an upper
 *                 bound; pqd_chunk_set_value_counts_dev lowers it to the non-null count once the
 *                 definition levels are decoded); out = val_out + value_base[i] * elem_size;
 *                 dict / dict_len as given.  PLAIN pages of a BYTE_ARRAY column, DELTA_LENGTH_BYTE_ARRAY and
 *                 DELTA_BYTE_ARRAY pages get encoding PQD_ENC_PLAIN_BYTE_ARRAY / _DELTA_LENGTH_BYTE_ARRAY /
 *                 _DELTA_BYTE_ARRAY, elem_size 0 and no `out`: their values go through pqd_delta_byte_array_dev
 *                 (a pqd_delta_bytes_desc takes data, data_len, num_values and encoding from this descriptor).
 *                 An encoding the column's type cannot have gives encoding 0: pqd_decode_pages_dev rejects it.
 * results[i].status: PQD_OK, or PQD_EOF / PQD_DECODE when a level section runs past the page.
 */
typedef struct pqd_column_meta {
  int32_t physical_type; /* enum pqd_physical_type; BOOLEAN = 0                                          */
  int32_t type_length;   /* FIXED_LEN_BYTE_ARRAY                                                         */
  int32_t max_def;       /* ColumnDescriptor.getMaxDefinitionLevel()                                     */
  int32_t max_rep;       /* ColumnDescriptor.getMaxRepetitionLevel()                                     */
} pqd_column_meta;
#define PQD_TYPE_BOOLEAN 0
typedef struct pqd_chunk_outputs {
  const uint8_t* bodies;        /* DEVICE: the uncompressed page bodies                                  */
  int32_t* rep_out;             /* DEVICE: int32 levels of the whole chunk (NULL when max_rep == 0)      */
  int32_t* def_out;             /* likewise                                                              */
  void* val_out;                /* DEVICE: dense non-null values of the whole chunk                      */
  const uint32_t* level_base;   /* DEVICE: per page, index of its first level in rep_out / def_out       */
  const uint32_t* value_base;   /* DEVICE: per page, index of its first value in val_out (an upper bound
                                   layout: level_base works when nulls leave gaps; see the set-counts call) */
  const void* dict;             /* DEVICE: dictionary entries (pqd_dict_info.entries) or NULL            */
  uint32_t dict_len;
  uint32_t reserved;
} pqd_chunk_outputs;
int pqd_chunk_sections_dev(const pqd_chunk_page* sections, uint32_t n_pages, const pqd_column_meta* column,
                           const pqd_chunk_outputs* outputs, pqd_page_desc* rep_descs, pqd_page_desc* def_descs,
                           pqd_page_desc* val_descs, pqd_page_result* results, void* stream);
/* Sets val_descs[i].num_values to the number of non-null values of page i, as the validity results
 * (pqd_def_levels_to_validity_dev: pqd_page_result.reserved) report it.  DEVICE arrays. */
int pqd_chunk_set_value_counts_dev(pqd_page_desc* val_descs, const pqd_page_result* validity_results, uint32_t n_pages,
                                   void* stream);

/* ---- page decompression --------------------------------------------------------------------------- */
/*
 * Replaces BytesInputDecompressor.decompress(bytes, uncompressedSize) as ColumnChunkPageReadStore calls it
 * for every page (parquet-hadoop/.../hadoop/ColumnChunkPageReadStore.java:151-181 V1 and dictionary pages,
 * :232-262 V2 where only the values behind the level bytes are compressed; CodecFactory.java:176-210), on
 * the DEVICE and for many pages per call.  Codecs (numbered as parquet-format's CompressionCodec, which
 * CompressionCodecName.java:26-33 mirrors): SNAPPY = the raw Snappy format (SnappyDecompressor.java:27-35
 * -> org.xerial.snappy.Snappy.uncompress), LZ4_RAW = the LZ4 block format (Lz4RawDecompressor.java:29-49
 * -> io.airlift.compress.lz4.Lz4Decompressor), ZSTD = Zstandard frames without dictionaries, RFC 8878
 * (ZstdDecompressorStream.java:30-60 -> zstd-jni's ZstdInputStreamNoFinalizer; the content checksum is not
 * verified), GZIP = one or more gzip members, RFC 1952 around RFC 1951 deflate data (CodecFactory.java:165-217
 * -> org.apache.hadoop.io.compress.GzipCodec, CompressionCodecName.java:28; every member's CRC-32 and length
 * are verified), UNCOMPRESSED = a copy.  LZO, BROTLI and the Hadoop-framed LZ4 are not decoded here.  A page must
 * produce exactly dst_len bytes (the reference reads uncompressed_page_size bytes from the codec's stream):
 * PQD_DECODE for a malformed stream, one that would write past dst_len, or one that ends short of it ("could
 * not decompress page"); result.values_written = bytes produced, bytes_consumed = source bytes read (GZIP: 0
 * unless the page decodes).  desc.codec is the call's `codec` or PQD_CODEC_UNCOMPRESSED (stored parts of the
 * same batch); a descriptor that names any other codec gets PQD_BAD_ARG.
 */
enum pqd_codec {
  PQD_CODEC_UNCOMPRESSED = 0, PQD_CODEC_SNAPPY = 1, PQD_CODEC_GZIP = 2, PQD_CODEC_LZO = 3, PQD_CODEC_BROTLI = 4,
  PQD_CODEC_LZ4 = 5, PQD_CODEC_ZSTD = 6, PQD_CODEC_LZ4_RAW = 7
};
typedef struct pqd_decompress_desc {
  const uint8_t* src; /* DEVICE: compressed bytes                                                          */
  uint8_t* dst;       /* DEVICE: dst_len bytes                                                             */
  uint32_t src_len;   /* below 2^31                                                                        */
  uint32_t dst_len;   /* the uncompressed size the page header announces                                   */
  uint32_t codec;     /* enum pqd_codec                                                                    */
  uint32_t reserved;
} pqd_decompress_desc;
int pqd_decompress_pages_dev(const pqd_decompress_desc* descs, pqd_page_result* results, uint32_t n_pages, int codec,
                             void* stream);
/* Whether pqd_decompress_pages_dev handles `codec` (UNCOMPRESSED, SNAPPY, GZIP, LZ4_RAW, ZSTD). */
int pqd_codec_supported(int codec);
/*
 * All pages of a walked chunk: `pages` is the HOST table of pqd_chunk_walk, `chunk` the chunk's bytes on
 * the DEVICE.  Every page body is laid out uncompressed in `bodies` (DEVICE, bodies_capacity bytes; the
 * sum of the uncompressed sizes, each rounded up to 16, is enough), V2 level bytes copied as they are.
 * uncompressed_pages (HOST, n_pages entries) receives the same table with body_offset = the page's place
 * in `bodies`, compressed_size = uncompressed_size and is_compressed = 0: the table
 * pqd_chunk_sections_dev and pqd_dict_create take.  descs / results: DEVICE scratch of 2 * n_pages
 * entries each; results[2 * i] and results[2 * i + 1] are page i's (a page has at most two parts).
 * PQD_BAD_ARG: unsupported codec, bodies too small.
 */
int pqd_chunk_decompress_dev(const uint8_t* chunk, const pqd_chunk_page* pages, uint32_t n_pages, int codec,
                             uint8_t* bodies, size_t bodies_capacity, pqd_chunk_page* uncompressed_pages,
                             pqd_decompress_desc* descs, pqd_page_result* results, void* stream);
/* One page body in HOST memory -> dst_len bytes in HOST memory.  Thread-safe like the other host entries. */
int pqd_decompress_page(int codec, const uint8_t* src, uint32_t src_len, uint8_t* dst, uint32_t dst_len,
                        pqd_page_result* result);

#ifdef __cplusplus
}
#endif
#endif /* PQD_H */
