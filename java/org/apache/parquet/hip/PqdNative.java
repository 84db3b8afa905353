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
      ENC_BIT_PACKED = 5;
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
