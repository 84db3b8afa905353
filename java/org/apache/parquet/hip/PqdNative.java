/*
 * FFM (java.lang.foreign, JDK 22+) binding of libpqdecode_hip.so -- the C ABI declared in include/pqd.h.
 * NOT COMPILED in this repository (the build image has no JDK); it is the binding a parquet-java
 * maintainer would add next to parquet-plugins/parquet-encoding-vector.
 */
package org.apache.parquet.hip;

import java.lang.foreign.Arena;
import java.lang.foreign.FunctionDescriptor;
import java.lang.foreign.Linker;
import java.lang.foreign.MemoryLayout;
import java.lang.foreign.MemorySegment;
import java.lang.foreign.StructLayout;
import java.lang.foreign.SymbolLookup;
import java.lang.invoke.MethodHandle;

import static java.lang.foreign.ValueLayout.ADDRESS;
import static java.lang.foreign.ValueLayout.JAVA_INT;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

public final class PqdNative {
  public static final int OK = 0, EOF = 1, BAD_ARG = 2, PAST_END = 3, DECODE = 4, DICT_OOB = 5, HIP_ERROR = 6, NO_DEVICE = 7;
  public static final int ENC_RLE = 1, ENC_RLE_DICTIONARY = 2, ENC_DELTA_BINARY_PACKED = 3, ENC_BYTE_STREAM_SPLIT = 4,
      ENC_BIT_PACKED = 5, ENC_PLAIN = 6, ENC_PLAIN_BOOLEAN = 7, ENC_DELTA_LENGTH_BYTE_ARRAY = 8, ENC_DELTA_BYTE_ARRAY = 9,
      ENC_PLAIN_BYTE_ARRAY = 10;
  /** enum pqd_codec: parquet-format's CompressionCodec numbers. */
  public static final int CODEC_UNCOMPRESSED = 0, CODEC_SNAPPY = 1, CODEC_GZIP = 2, CODEC_ZSTD = 6, CODEC_LZ4_RAW = 7;
  public static final int F_LENGTH_PREFIX = 1, F_WIDTH_PREFIX = 2;

  /** struct pqd_page_desc (40 bytes). */
  public static final StructLayout PAGE_DESC = MemoryLayout.structLayout(
      ADDRESS.withName("data"), ADDRESS.withName("out"), ADDRESS.withName("dict"),
      JAVA_INT.withName("data_len"), JAVA_INT.withName("num_values"), JAVA_INT.withName("dict_len"),
      JAVA_INT.withName("encoding_elem_width_flags")); // uint8 encoding, elem_size, bit_width, flags
  /** struct pqd_page_result (16 bytes). */
  public static final StructLayout PAGE_RESULT = MemoryLayout.structLayout(
      JAVA_INT.withName("status"), JAVA_INT.withName("values_written"), JAVA_INT.withName("bytes_consumed"),
      JAVA_INT.withName("reserved"));

  /** struct pqd_chunk_page (48 bytes): one page of a walked column chunk. */
  public static final StructLayout CHUNK_PAGE = MemoryLayout.structLayout(
      JAVA_INT.withName("header_offset"), JAVA_INT.withName("body_offset"), JAVA_INT.withName("compressed_size"),
      JAVA_INT.withName("uncompressed_size"), JAVA_INT.withName("num_values"), JAVA_INT.withName("num_nulls"),
      JAVA_INT.withName("num_rows"), JAVA_INT.withName("rep_levels_len"), JAVA_INT.withName("def_levels_len"),
      JAVA_INT.withName("crc"),
      JAVA_INT.withName("type_encoding_def_rep"),     // uint8 type, encoding, def_encoding, rep_encoding
      JAVA_INT.withName("compressed_crc_reserved"));  // uint8 is_compressed, has_crc, reserved[2]
  /** struct pqd_column_meta (16 bytes). */
  public static final StructLayout COLUMN_META = MemoryLayout.structLayout(
      JAVA_INT.withName("physical_type"), JAVA_INT.withName("type_length"), JAVA_INT.withName("max_def"), JAVA_INT.withName("max_rep"));

  private static final Linker LINKER = Linker.nativeLinker();
  private static final SymbolLookup LIB =
      SymbolLookup.libraryLookup(System.getProperty("parquet.hip.library", "libpqdecode_hip.so"), Arena.global());

  private static MethodHandle h(String name, FunctionDescriptor fd, Linker.Option... opts) {
    return LINKER.downcallHandle(LIB.find(name).orElseThrow(() -> new UnsatisfiedLinkError(name)), fd, opts);
  }

  // Heap int[] / byte[] can be handed to a downcall only as DIRECT arguments of a handle linked with
  // Linker.Option.critical(true) (MemorySegment.ofArray(...), the FFM counterpart of JNI
  // GetPrimitiveArrayCritical).  The three entry points below take every buffer as an argument of its
  // own for that reason; they block until the device has answered (tens of microseconds per page).
  static final MethodHandle UNPACK_I32 = h("pqd_unpack_i32",
      FunctionDescriptor.of(JAVA_INT, JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS, JAVA_LONG), Linker.Option.critical(true));
  static final MethodHandle UNPACK_I64 = h("pqd_unpack_i64",
      FunctionDescriptor.of(JAVA_INT, JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS, JAVA_LONG), Linker.Option.critical(true));
  // pqd_decode_page_flat(encoding, elem_size, bit_width, flags, data, data_len, num_values, dict, dict_len, out, result)
  static final MethodHandle DECODE_PAGE_FLAT = h("pqd_decode_page_flat",
      FunctionDescriptor.of(JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, JAVA_INT,
          ADDRESS, ADDRESS), Linker.Option.critical(true));
  // Pointers INSIDE a struct cannot be heap arrays, and calls that allocate or copy large buffers
  // are not short: these handles are linked without `critical` and take native / direct-buffer memory.
  static final MethodHandle DECODE_PAGE = h("pqd_decode_page", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS));
  static final MethodHandle DECODE_PAGES = h("pqd_decode_pages",
      FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS));
  // dictionaries: pqd_dict_create(physical_type, type_length, page, page_len, n_entries, &dict) replaces
  // Encoding.PLAIN.initDictionary; pqd_decode_binary_page replaces the readBytes() loop of a
  // DictionaryValuesReader over a PlainBinaryDictionary (offsets[n+1] + bytes come back)
  static final MethodHandle DICT_CREATE = h("pqd_dict_create",
      FunctionDescriptor.of(JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS, JAVA_LONG, JAVA_INT, ADDRESS));
  static final MethodHandle DICT_DESTROY = h("pqd_dict_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
  static final MethodHandle DECODE_BINARY_PAGE = h("pqd_decode_binary_page",
      FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS));
  // ---- column chunks, page decompression, BYTE_ARRAY value pages, levels (host entries: buffers in native memory)
  // pqd_chunk_walk(chunk, chunk_len, value_count, verify_crc, pages, pages_capacity, &n_pages) replaces
  // ParquetFileReader.Chunk.readAllPages up to decompression
  static final MethodHandle CHUNK_WALK = h("pqd_chunk_walk",
      FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG, JAVA_LONG, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS));
  static final MethodHandle CRC32 = h("pqd_crc32", FunctionDescriptor.of(JAVA_INT, JAVA_INT, ADDRESS, JAVA_LONG));
  // pqd_decompress_page(codec, src, src_len, dst, dst_len, result) replaces BytesInputDecompressor.decompress
  static final MethodHandle DECOMPRESS_PAGE = h("pqd_decompress_page",
      FunctionDescriptor.of(JAVA_INT, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS));
  static final MethodHandle CODEC_SUPPORTED = h("pqd_codec_supported", FunctionDescriptor.of(JAVA_INT, JAVA_INT));
  // pqd_decode_delta_byte_array_page(encoding, page, page_len, num_values, previous, previous_len, offsets, bytes,
  // bytes_capacity, result, &bytes_needed) replaces the readBytes() loops of BinaryPlainValuesReader,
  // DeltaLengthByteArrayValuesReader and DeltaByteArrayReader (ENC_PLAIN_BYTE_ARRAY / 8 / 9)
  static final MethodHandle DECODE_BYTE_ARRAY_PAGE = h("pqd_decode_delta_byte_array_page",
      FunctionDescriptor.of(JAVA_INT, JAVA_INT, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS, ADDRESS));
  // ---- the batched DEVICE entries: descriptor and result arrays in device memory, one launch set per call
  static final MethodHandle DEV_ALLOC = h("pqd_dev_alloc", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG));
  static final MethodHandle DEV_FREE = h("pqd_dev_free", FunctionDescriptor.of(JAVA_INT, ADDRESS));
  static final MethodHandle HOST_ALLOC_PINNED = h("pqd_host_alloc_pinned", FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_LONG));
  static final MethodHandle HOST_FREE_PINNED = h("pqd_host_free_pinned", FunctionDescriptor.of(JAVA_INT, ADDRESS));
  static final MethodHandle MEMCPY_H2D = h("pqd_memcpy_h2d", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS));
  static final MethodHandle MEMCPY_D2H = h("pqd_memcpy_d2h", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_LONG, ADDRESS));
  static final MethodHandle STREAM_CREATE = h("pqd_stream_create", FunctionDescriptor.of(JAVA_INT, ADDRESS));
  static final MethodHandle STREAM_DESTROY = h("pqd_stream_destroy", FunctionDescriptor.of(JAVA_INT, ADDRESS));
  static final MethodHandle STREAM_SYNC = h("pqd_stream_sync", FunctionDescriptor.of(JAVA_INT, ADDRESS));
  // (descs, results, n_pages, encoding, elem_size, stream)
  static final MethodHandle DECODE_PAGES_DEV = h("pqd_decode_pages_dev",
      FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, JAVA_INT, ADDRESS));
  // (dict, pages, bins, page_results, bin_results, n_pages, stream): dictionary pages of a BINARY column -> offsets + bytes
  static final MethodHandle DECODE_BINARY_PAGES_DEV = h("pqd_decode_binary_pages_dev",
      FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, JAVA_INT, ADDRESS));
  static final MethodHandle DICT_GET = h("pqd_dict_get", FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS));
  // (descs, results, n_pages, stream): PLAIN BYTE_ARRAY / DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY pages
  static final MethodHandle BYTE_ARRAY_PAGES_DEV = h("pqd_delta_byte_array_dev",
      FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS));
  static final MethodHandle BINARY_MATERIALIZE_DEV = h("pqd_binary_materialize_dev",
      FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS));
  // levels -> validity / null scatter / list offsets (ColumnReaderBase.checkRead, :642-668)
  static final MethodHandle DEF_LEVELS_TO_VALIDITY_DEV = h("pqd_def_levels_to_validity_dev",
      FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS));
  static final MethodHandle SCATTER_NULLS_DEV = h("pqd_scatter_nulls_dev",
      FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS));
  static final MethodHandle REP_LEVELS_TO_OFFSETS_DEV = h("pqd_rep_levels_to_offsets_dev",
      FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS));
  // (sections, n_pages, &column_meta, &outputs, rep_descs, def_descs, val_descs, results, stream)
  static final MethodHandle CHUNK_SECTIONS_DEV = h("pqd_chunk_sections_dev",
      FunctionDescriptor.of(JAVA_INT, ADDRESS, JAVA_INT, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
  static final MethodHandle CHUNK_SET_VALUE_COUNTS_DEV = h("pqd_chunk_set_value_counts_dev",
      FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, ADDRESS));
  // (descs, results, n_pages, codec, stream)
  static final MethodHandle DECOMPRESS_PAGES_DEV = h("pqd_decompress_pages_dev",
      FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS));
  // (chunk, pages, n_pages, codec, bodies, bodies_capacity, uncompressed_pages, descs, results, stream)
  static final MethodHandle CHUNK_DECOMPRESS_DEV = h("pqd_chunk_decompress_dev",
      FunctionDescriptor.of(JAVA_INT, ADDRESS, ADDRESS, JAVA_INT, JAVA_INT, ADDRESS, JAVA_LONG, ADDRESS, ADDRESS, ADDRESS, ADDRESS));
  // gives back the calling thread's stream and staging buffers (executors that retire threads)
  static final MethodHandle THREAD_RELEASE = h("pqd_thread_release", FunctionDescriptor.of(JAVA_INT));
  static final MethodHandle DEVICE_COUNT = h("pqd_device_count", FunctionDescriptor.of(JAVA_INT));
  static final MethodHandle INIT = h("pqd_init", FunctionDescriptor.of(JAVA_INT, JAVA_INT));
  static final MethodHandle LAST_ERROR = h("pqd_last_error", FunctionDescriptor.of(ADDRESS));

  private PqdNative() {}

  public static int deviceCount() {
    try {
      return (int) DEVICE_COUNT.invokeExact();
    } catch (Throwable t) {
      throw new IllegalStateException(t);
    }
  }

  static String lastError() {
    try {
      MemorySegment s = (MemorySegment) LAST_ERROR.invokeExact();
      return s.reinterpret(4096).getString(0);
    } catch (Throwable t) {
      return "";
    }
  }

  /** Maps a library status to the exception the Java decoder would have thrown. */
  static void check(int status, String what) throws java.io.IOException {
    switch (status) {
      case OK: return;
      case EOF: throw new java.io.EOFException(what);
      case BAD_ARG: throw new IllegalArgumentException(what + ": " + lastError());
      case PAST_END: throw new IllegalArgumentException("Reading past RLE/BitPacking stream.");
      case DECODE:
      case DICT_OOB: throw new org.apache.parquet.io.ParquetDecodingException(what + ": " + lastError());
      default: throw new IllegalStateException(what + ": " + lastError());
    }
  }

  /**
   * BytesInputDecompressor.decompress(bytes, uncompressedSize) for one page: `src` and `dst` are native (or direct
   * buffer) segments; the page must produce exactly dst.byteSize() bytes.
   */
  public static void decompressPage(int codec, MemorySegment src, MemorySegment dst) throws java.io.IOException {
    try (Arena a = Arena.ofConfined()) {
      MemorySegment result = a.allocate(PAGE_RESULT);
      int st = (int) DECOMPRESS_PAGE.invokeExact(codec, src, (int) src.byteSize(), dst, (int) dst.byteSize(), result);
      check(st, "pqd_decompress_page");
      if (result.get(JAVA_INT, 0) != OK) throw new java.io.IOException("could not decompress page");
    } catch (java.io.IOException | RuntimeException e) {
      throw e;
    } catch (Throwable t) {
      throw new RuntimeException(t);
    }
  }

  /**
   * The page table of one column chunk (ParquetFileReader.Chunk.readAllPages up to decompression): returns the
   * number of pages written to `pages` (an array of CHUNK_PAGE structs in native memory).
   */
  public static int chunkWalk(MemorySegment chunk, long valueCount, boolean verifyCrc, MemorySegment pages) throws java.io.IOException {
    try (Arena a = Arena.ofConfined()) {
      MemorySegment n = a.allocate(JAVA_INT);
      int capacity = (int) (pages.byteSize() / CHUNK_PAGE.byteSize());
      int st = (int) CHUNK_WALK.invokeExact(chunk, chunk.byteSize(), valueCount, verifyCrc ? 1 : 0, pages, capacity, n);
      check(st, "pqd_chunk_walk");
      return n.get(JAVA_INT, 0);
    } catch (java.io.IOException | RuntimeException e) {
      throw e;
    } catch (Throwable t) {
      throw new RuntimeException(t);
    }
  }

  /** One step of the per-step packer interface: `count` values of `bitWidth` bits from in[inPos...] to out[outPos...]. */
  public static void unpackI32(int bitWidth, byte[] in, int inPos, int[] out, int outPos, int count) {
    unpackI32(bitWidth, MemorySegment.ofArray(in), inPos, out, outPos, count);
  }

  public static void unpackI32(int bitWidth, java.nio.ByteBuffer in, int inPos, int[] out, int outPos, int count) {
    unpackI32(bitWidth, MemorySegment.ofBuffer(in), inPos, out, outPos, count);
  }

  private static void unpackI32(int bitWidth, MemorySegment in, int inPos, int[] out, int outPos, int count) {
    if (count == 0) return;
    long bytes = (long) count * bitWidth / 8;
    try {
      int st = (int) UNPACK_I32.invokeExact(bitWidth, in.asSlice(inPos, bytes), bytes,
          MemorySegment.ofArray(out).asSlice(4L * outPos, 4L * count), (long) count);
      if (st != OK) throw new RuntimeException("pqd_unpack_i32: " + lastError());
    } catch (RuntimeException e) {
      throw e;
    } catch (Throwable t) {
      throw new RuntimeException(t);
    }
  }
}
