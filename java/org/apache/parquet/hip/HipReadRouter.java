/*
 * Drop-in for org.apache.parquet.column.values.bitpacking.ParquetReadRouter
 * (parquet-plugins/parquet-encoding-vector/.../ParquetReadRouter.java:57-116): same static methods,
 * same arguments, same stream consumption (exactly currentCount * bitWidth / 8 bytes), EOFException
 * when the stream is shorter.  NOT COMPILED here (no JDK in the build image).
 */
package org.apache.parquet.hip;

import java.io.IOException;
import java.lang.foreign.MemorySegment;
import java.nio.ByteBuffer;
import org.apache.parquet.bytes.ByteBufferInputStream;

public final class HipReadRouter {
  private HipReadRouter() {}

  public static void read(int bitWidth, ByteBufferInputStream in, int currentCount, int[] currentBuffer)
      throws IOException {
    readBatch(bitWidth, in, currentCount, currentBuffer);
  }

  public static void readBatchUsing512Vector(int bitWidth, ByteBufferInputStream in, int currentCount,
      int[] currentBuffer) throws IOException {
    readBatch(bitWidth, in, currentCount, currentBuffer);
  }

  public static void readBatch(int bitWidth, ByteBufferInputStream in, int currentCount, int[] currentBuffer)
      throws IOException {
    int bytes = currentCount / 8 * bitWidth; // ParquetReadRouter.java:77-79
    ByteBuffer buf = in.slice(bytes);        // throws EOFException when short, like the reference
    MemorySegment src = buf.isDirect() ? MemorySegment.ofBuffer(buf)
        : MemorySegment.ofArray(buf.array()).asSlice(buf.arrayOffset() + buf.position(), bytes);
    int status;
    try {
      status = (int) PqdNative.UNPACK_I32.invokeExact(bitWidth, src, (long) bytes,
          MemorySegment.ofArray(currentBuffer), (long) currentCount);
    } catch (Throwable t) {
      throw new IllegalStateException(t);
    }
    PqdNative.check(status, "pqd_unpack_i32");
  }
}
