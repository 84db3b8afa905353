/*
 * Drop-in for org.apache.parquet.column.values.dictionary.DictionaryValuesReader
 * (parquet-column/.../values/dictionary/DictionaryValuesReader.java:44-127) for INT32 / INT64 / FLOAT /
 * DOUBLE dictionaries: initFromPage decodes the whole page on the device (eagerly, like the
 * reference's delta and byte-stream-split readers already do) and the read methods serve the decoded
 * array.  Encoding.getDictionaryBasedValuesReader (Encoding.java:306-323) would return this class when
 * PqdNative.deviceCount() > 0.  NOT COMPILED here (no JDK in the build image).
 */
package org.apache.parquet.hip;

import java.io.IOException;
import java.lang.foreign.Arena;
import java.lang.foreign.MemorySegment;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import org.apache.parquet.bytes.ByteBufferInputStream;
import org.apache.parquet.column.values.ValuesReader;
import org.apache.parquet.io.ParquetDecodingException;

public class HipDictionaryValuesReader extends ValuesReader {
  private final MemorySegment dictionary; // little-endian fixed-width entries, off-heap
  private final int dictLen, elemSize;
  private ByteBuffer decoded;
  private int written, next;
  private int failure = PqdNative.OK;

  /** dictionaryPage: the PLAIN dictionary page bytes (PlainValuesDictionary.java:148-157 = LE memcpy). */
  public HipDictionaryValuesReader(ByteBuffer dictionaryPage, int dictionarySize, int elemSize, Arena arena) {
    this.dictionary = arena.allocate((long) dictionarySize * elemSize, 16).copyFrom(MemorySegment.ofBuffer(dictionaryPage));
    this.dictLen = dictionarySize;
    this.elemSize = elemSize;
  }

  @Override
  public void initFromPage(int valueCount, ByteBufferInputStream stream) throws IOException {
    ByteBuffer page = stream.slice(stream.available()); // the id stream runs to the end of the page (:50-55)
    decoded = ByteBuffer.allocateDirect(Math.max(valueCount, 1) * elemSize).order(ByteOrder.LITTLE_ENDIAN);
    try (Arena a = Arena.ofConfined()) {
      MemorySegment d = a.allocate(PqdNative.PAGE_DESC), r = a.allocate(PqdNative.PAGE_RESULT);
      MemorySegment data = page.isDirect() ? MemorySegment.ofBuffer(page)
          : a.allocate(Math.max(page.remaining(), 1)).copyFrom(MemorySegment.ofBuffer(page));
      d.set(java.lang.foreign.ValueLayout.ADDRESS, 0, data);
      d.set(java.lang.foreign.ValueLayout.ADDRESS, 8, MemorySegment.ofBuffer(decoded));
      d.set(java.lang.foreign.ValueLayout.ADDRESS, 16, dictionary);
      d.set(java.lang.foreign.ValueLayout.JAVA_INT, 24, page.remaining());
      d.set(java.lang.foreign.ValueLayout.JAVA_INT, 28, valueCount);
      d.set(java.lang.foreign.ValueLayout.JAVA_INT, 32, dictLen);
      d.set(java.lang.foreign.ValueLayout.JAVA_INT, 36, PqdNative.ENC_RLE_DICTIONARY | (elemSize << 8));
      int st;
      try {
        st = (int) PqdNative.DECODE_PAGE.invokeExact(d, r);
      } catch (Throwable t) {
        throw new IllegalStateException(t);
      }
      if (st != PqdNative.OK) PqdNative.check(st, "pqd_decode_page");
      // per-page status: values before the failure are served, the failing read throws (as in Java)
      failure = r.get(java.lang.foreign.ValueLayout.JAVA_INT, 0);
      written = r.get(java.lang.foreign.ValueLayout.JAVA_INT, 4);
    }
    next = 0;
  }

  private int index() {
    if (next >= written) {
      if (failure == PqdNative.OK) throw new ParquetDecodingException("no more values in this page");
      try {
        PqdNative.check(failure, "dictionary page");
      } catch (IOException e) {
        throw new ParquetDecodingException(e);
      }
    }
    return next++;
  }

  @Override public int readInteger() { return decoded.getInt(index() * 4); }
  @Override public long readLong() { return decoded.getLong(index() * 8); }
  @Override public float readFloat() { return decoded.getFloat(index() * 4); }
  @Override public double readDouble() { return decoded.getDouble(index() * 8); }
  @Override public void skip() { index(); }
  @Override public void skip(int n) { for (int i = 0; i < n; ++i) index(); }
}
