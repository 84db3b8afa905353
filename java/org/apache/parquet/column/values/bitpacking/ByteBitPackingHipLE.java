/*
 * Sketch (not compiled in this repository: the build image has no JDK).
 *
 * The plugin class that Packer.LITTLE_ENDIAN.newBytePackerVector(width) looks up by reflection
 * (parquet-encoding/.../Packer.java:68-111: Class.forName(...).getField("factory")): the same shape as
 * ByteBitPacking512VectorLE (parquet-plugins/parquet-encoding-vector/.../ByteBitPacking512VectorLE.java:37-79),
 * 33 stateless packers, one per width.  To be picked up by an unmodified Packer it has to be published under
 * the name ByteBitPacking512VectorLE; under its own name it needs the one-line change in Packer.
 */
package org.apache.parquet.column.values.bitpacking;

import java.nio.ByteBuffer;
import org.apache.parquet.hip.PqdNative;

public abstract class ByteBitPackingHipLE {
  private static final BytePacker[] packers = new BytePacker[33];

  static {
    for (int w = 0; w <= 32; ++w) packers[w] = new HipPacker(w);
  }

  public static final BytePackerFactory factory = new BytePackerFactory() {
    @Override
    public BytePacker newBytePacker(int bitWidth) {
      return packers[bitWidth];
    }
  };

  private static final class HipPacker extends BytePacker {
    HipPacker(int bitWidth) {
      super(bitWidth);
    }

    /** values per unpackValuesUsingVector step, as the AVX-512 plugin reports them */
    @Override
    public int getUnpackCount() {
      int w = getBitWidth();
      return w == 0 ? 0 : w == 8 ? 64 : w < 16 ? 32 : 16;
    }

    @Override
    public void unpackValuesUsingVector(byte[] in, int inPos, int[] out, int outPos) {
      PqdNative.unpackI32(getBitWidth(), in, inPos, out, outPos, getUnpackCount());
    }

    @Override
    public void unpackValuesUsingVector(ByteBuffer in, int inPos, int[] out, int outPos) {
      PqdNative.unpackI32(getBitWidth(), in, inPos, out, outPos, getUnpackCount());
    }

    @Override
    public void unpack8Values(ByteBuffer in, int inPos, int[] out, int outPos) {
      PqdNative.unpackI32(getBitWidth(), in, inPos, out, outPos, 8);
    }

    @Override
    public void unpack32Values(ByteBuffer in, int inPos, int[] out, int outPos) {
      PqdNative.unpackI32(getBitWidth(), in, inPos, out, outPos, 32);
    }

    // the decode library packs nothing; the AVX-512 plugin throws from these too (ByteBitPacking512VectorLE.java:3229-3232)
    @Override
    public void pack8Values(int[] in, int inPos, byte[] out, int outPos) {
      throw new RuntimeException("pack8Values is not supported by the HIP plugin");
    }

    @Override
    public void pack32Values(int[] in, int inPos, byte[] out, int outPos) {
      throw new RuntimeException("pack32Values is not supported by the HIP plugin");
    }
  }
}
