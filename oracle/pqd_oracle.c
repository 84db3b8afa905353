/*
 * pqd_oracle.c -- CPU restatement of parquet-java's column-page value decoders.
 *
 * TEST INFRASTRUCTURE ONLY (see pqd_oracle.h).  Never linked into, loaded by or
 * called from the shipped HIP library.
 *
 * Reference paths are relative to the apache/parquet-java source tree:
 *   GEN  = parquet-generator/src/main/java/org/apache/parquet/encoding/bitpacking
 *   COL  = parquet-column/src/main/java/org/apache/parquet/column
 *   CMN  = parquet-common/src/main/java/org/apache/parquet
 *   VEC  = parquet-plugins/parquet-encoding-vector/src/main/java/org/apache/parquet/column/values/bitpacking
 */
#include "pqd_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* Bit packing.                                                               */
/* The generated Java classes (ByteBitPackingLE/BE, ByteBitPackingForLongLE/BE)*/
/* are not in the reference tree; their arithmetic is defined by the generator */
/* GEN/ByteBasedBitPackingGenerator.java.  What follows evaluates the          */
/* generator's shift/mask rule at run time instead of emitting source text.    */
/* ------------------------------------------------------------------------- */

typedef struct {
  int shift;
  uint64_t mask;
} shift_mask;

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* GEN/ByteBasedBitPackingGenerator.java:310-323 (genMask) */
static uint64_t gen_mask(int width, int is_long) {
  int max_width = is_long ? 64 : 32;
  if (width >= max_width) return is_long ? ~(uint64_t)0 : (uint64_t)0xFFFFFFFFu;
  return (((uint64_t)1) << width) - 1;
}

/* GEN/ByteBasedBitPackingGenerator.java:136-182 (getShift) */
static shift_mask get_shift(int w, int is_long, int msb_first, int byte_index, int value_index) {
  int value_start = value_index * w - 8 * byte_index;
  int value_end = (value_index + 1) * w - 8 * (byte_index + 1);
  int vs_wanted, ve_wanted, bs_wanted, be_wanted, shift, width_wanted;
  if (msb_first) {
    vs_wanted = value_start < 0 ? w - 1 + value_start : w - 1;
    ve_wanted = value_end > 0 ? value_end : 0;
    bs_wanted = value_start < 0 ? 8 : 7 - value_start;
    be_wanted = value_end > 0 ? 0 : -value_end;
    shift = ve_wanted - be_wanted;
    width_wanted = imin(7, bs_wanted) - imin(7, be_wanted) + 1;
  } else {
    vs_wanted = w - 1 - (value_end > 0 ? value_end : 0);
    ve_wanted = w - 1 - (value_start < 0 ? w - 1 + value_start : w - 1);
    bs_wanted = 7 - (value_end > 0 ? 0 : -value_end);
    be_wanted = 7 - (value_start < 0 ? 8 : 7 - value_start);
    shift = vs_wanted - bs_wanted;
    width_wanted = imax(0, bs_wanted) - imax(0, be_wanted) + 1;
  }
  (void)vs_wanted;
  (void)ve_wanted;
  shift_mask sm;
  sm.shift = shift;
  sm.mask = gen_mask(width_wanted + imax(0, shift), is_long);
  return sm;
}

/* The generator bakes one (byte, shift, mask) term list per value into straight-line Java.
 * The same lists are built once here, when the library is loaded, so the unpack loops below do
 * the work the generated code does at run time and nothing more. */
typedef struct {
  uint8_t byte;
  int8_t shift;
  uint64_t mask;
} unpack_term;
typedef struct {
  uint8_t n[8];
  unpack_term t[8][9];
} unpack_plan;
static unpack_plan plan32[2][33];
static unpack_plan plan64[2][65];

static void build_plan(unpack_plan* p, int w, int is_long, int msb_first) {
  memset(p, 0, sizeof *p);
  for (int v = 0; v < 8; ++v) {
    int start = v * w / 8;
    int end = ((v + 1) * w + 7) / 8;
    for (int b = start; b < end; ++b) {
      shift_mask sm = get_shift(w, is_long, msb_first, b, v);
      unpack_term* t = &p->t[v][p->n[v]++];
      t->byte = (uint8_t)b;
      t->shift = (int8_t)sm.shift;
      t->mask = sm.mask;
    }
  }
}

__attribute__((constructor)) static void build_plans(void) {
  for (int m = 0; m < 2; ++m) {
    for (int w = 1; w <= 32; ++w) build_plan(&plan32[m][w], w, 0, m);
    for (int w = 1; w <= 64; ++w) build_plan(&plan64[m][w], w, 1, m);
  }
}

/* GEN/ByteBasedBitPackingGenerator.java:258-308 (generateUnpack), int flavour.
 * Java: out[v] = OR over bytes of ((((int)in[b]) >> / << s) & mask); the byte is
 * sign-extended, the mask removes the extension. */
void orc_unpack8_32(int msb_first, int w, const uint8_t* in, int32_t* out) {
  if (w <= 0) return; /* :267 -- width 0 writes nothing */
  const unpack_plan* p = &plan32[msb_first ? 1 : 0][w];
  for (int v = 0; v < 8; ++v) {
    uint32_t acc = 0;
    for (int k = 0; k < p->n[v]; ++k) {
      const unpack_term* t = &p->t[v][k];
      int32_t sx = (int32_t)(int8_t)in[t->byte];
      uint32_t x;
      if (t->shift < 0)
        x = (uint32_t)(sx >> (-t->shift)); /* arithmetic, as Java >> */
      else
        x = ((uint32_t)sx) << t->shift;
      acc |= x & (uint32_t)t->mask;
    }
    out[v] = (int32_t)acc;
  }
}

/* same, long flavour (ByteBitPackingForLongLE/BE, widths 0..64) */
void orc_unpack8_64(int msb_first, int w, const uint8_t* in, int64_t* out) {
  if (w <= 0) return;
  const unpack_plan* p = &plan64[msb_first ? 1 : 0][w];
  for (int v = 0; v < 8; ++v) {
    uint64_t acc = 0;
    for (int k = 0; k < p->n[v]; ++k) {
      const unpack_term* t = &p->t[v][k];
      int64_t sx = (int64_t)(int8_t)in[t->byte];
      uint64_t x;
      if (t->shift < 0)
        x = (uint64_t)(sx >> (-t->shift));
      else
        x = ((uint64_t)sx) << t->shift;
      acc |= x & t->mask;
    }
    out[v] = (int64_t)acc;
  }
}

/* GEN/ByteBasedBitPackingGenerator.java:224-256 (generatePack) */
void orc_pack8_32(int msb_first, int w, const int32_t* in, uint8_t* out) {
  uint32_t mask = (uint32_t)gen_mask(w, 0);
  for (int b = 0; b < w; ++b) {
    int start = (b * 8) / w;
    int end = ((b + 1) * 8 + w - 1) / w;
    uint32_t acc = 0;
    for (int v = start; v < end; ++v) {
      shift_mask sm = get_shift(w, 0, msb_first, b, v);
      uint32_t x = ((uint32_t)in[v]) & mask;
      if (sm.shift > 0)
        acc |= x >> sm.shift; /* >>> */
      else if (sm.shift < 0)
        acc |= x << (-sm.shift);
      else
        acc |= x;
    }
    out[b] = (uint8_t)(acc & 255u);
  }
}

void orc_pack8_64(int msb_first, int w, const int64_t* in, uint8_t* out) {
  uint64_t mask = gen_mask(w, 1);
  for (int b = 0; b < w; ++b) {
    int start = (b * 8) / w;
    int end = ((b + 1) * 8 + w - 1) / w;
    uint64_t acc = 0;
    for (int v = start; v < end; ++v) {
      shift_mask sm = get_shift(w, 1, msb_first, b, v);
      uint64_t x = ((uint64_t)in[v]) & mask;
      if (sm.shift > 0)
        acc |= x >> sm.shift;
      else if (sm.shift < 0)
        acc |= x << (-sm.shift);
      else
        acc |= x;
    }
    out[b] = (uint8_t)(acc & 255u);
  }
}

/* ------------------------------------------------------------------------- */
/* ParquetReadRouter                                                          */
/* ------------------------------------------------------------------------- */

/* VEC/ByteBitPacking512VectorLE.java: getUnpackCount() per width
 * (0 -> 0; 1-7, 9-15 -> 32; 8 -> 64; 16-32 -> 16). */
static int vector_unpack_count(int w) {
  if (w == 0) return 0;
  if (w == 8) return 64;
  if (w < 16) return 32;
  return 16;
}

/* VEC/ParquetReadRouter.java:69-104 (readBatchUsing512Vector).  One vector step
 * yields the same ints as unpackCount/8 scalar unpack8Values calls
 * (VECT/TestByteBitPacking512VectorLE.java:60-87), so the step is restated with
 * the scalar kernel; the two-loop structure and the -64 guard are kept. */
int orc_read_batch_i32(int w, const uint8_t* in, size_t in_len, int32_t* out, size_t n,
                       size_t* consumed) {
  if (consumed) *consumed = 0;
  if (w < 0 || w > 32 || (n & 7) != 0) return ORC_BAD_ARG;
  size_t total = n * (size_t)w / 8;
  if (in_len < total) return ORC_EOF; /* in.slice(totalByteCount), :79 */
  long total_vec = (long)total - 64;
  int step_values = vector_unpack_count(w);
  int step_bytes = step_values / 8 * w;
  size_t vi = 0;
  long bi = 0;
  if (step_bytes > 0) {
    for (; bi < total_vec; bi += step_bytes, vi += (size_t)step_values)
      for (int g = 0; g < step_values / 8; ++g)
        orc_unpack8_32(0, w, in + bi + (long)g * w, out + vi + 8 * (size_t)g);
  }
  if (w > 0) {
    for (; bi < (long)total; bi += w, vi += 8) orc_unpack8_32(0, w, in + bi, out + vi);
  }
  /* width 0: the Java loops execute no unpack step; the int[] keeps its zeros */
  if (w == 0) memset(out, 0, n * sizeof(int32_t));
  if (consumed) *consumed = total;
  return ORC_OK;
}

/* BytePackerForLong.unpack8Values applied to a whole run
 * (ENC/BytePackerForLong.java:72-105; caller shape DeltaBinaryPackingValuesReader.java:150-162) */
int orc_read_batch_i64(int w, const uint8_t* in, size_t in_len, int64_t* out, size_t n,
                       size_t* consumed) {
  if (consumed) *consumed = 0;
  if (w < 0 || w > 64 || (n & 7) != 0) return ORC_BAD_ARG;
  size_t total = n / 8 * (size_t)w;
  if (in_len < total) return ORC_EOF;
  if (w == 0) {
    memset(out, 0, n * sizeof(int64_t));
  } else {
    for (size_t g = 0; g < n / 8; ++g) orc_unpack8_64(0, w, in + g * (size_t)w, out + 8 * g);
  }
  if (consumed) *consumed = total;
  return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* BytesUtils                                                                 */
/* ------------------------------------------------------------------------- */

/* CMN/bytes/BytesUtils.java:202-211; Java int shifts use the low 5 bits of the count */
int orc_read_uvarint32(const uint8_t* in, size_t len, size_t* pos, uint32_t* v) {
  uint32_t value = 0;
  int i = 0;
  for (;;) {
    if (*pos >= len) return ORC_EOF;
    uint32_t b = in[(*pos)++];
    if (b & 0x80u) {
      value |= (b & 0x7Fu) << (i & 31);
      i += 7;
    } else {
      *v = value | (b << (i & 31));
      return ORC_OK;
    }
  }
}

/* CMN/bytes/BytesUtils.java:260-269 (readUnsignedVarLong): long shifts use the low 6 bits */
static int read_uvarlong(const uint8_t* in, size_t len, size_t* pos, uint64_t* v) {
  uint64_t value = 0;
  int i = 0;
  for (;;) {
    if (*pos >= len) return ORC_EOF;
    uint64_t b = in[(*pos)++];
    if (b & 0x80u) {
      value |= (b & 0x7Fu) << (i & 63);
      i += 7;
    } else {
      *v = value | (b << (i & 63));
      return ORC_OK;
    }
  }
}

/* CMN/bytes/BytesUtils.java:254-258 */
int orc_read_zigzag_varlong(const uint8_t* in, size_t len, size_t* pos, int64_t* v) {
  uint64_t raw;
  int st = read_uvarlong(in, len, pos, &raw);
  if (st) return st;
  int64_t sraw = (int64_t)raw;
  int64_t temp = ((int64_t)(((int64_t)(raw << 63)) >> 63) ^ sraw) >> 1;
  *v = temp ^ (int64_t)(raw & ((uint64_t)1 << 63));
  return ORC_OK;
}

static size_t put_uvarint32(uint32_t value, uint8_t* out) { /* BytesUtils.java:226-232 */
  size_t n = 0;
  while ((value & 0xFFFFFF80u) != 0) {
    out[n++] = (uint8_t)((value & 0x7F) | 0x80);
    value >>= 7;
  }
  out[n++] = (uint8_t)(value & 0x7F);
  return n;
}

static size_t put_uvarlong(uint64_t value, uint8_t* out) { /* BytesUtils.java:271-277 */
  size_t n = 0;
  while ((value & 0xFFFFFFFFFFFFFF80ull) != 0) {
    out[n++] = (uint8_t)((value & 0x7F) | 0x80);
    value >>= 7;
  }
  out[n++] = (uint8_t)(value & 0x7F);
  return n;
}

/* ------------------------------------------------------------------------- */
/* RLE / bit-packing hybrid decoder                                           */
/* COL/values/rle/RunLengthBitPackingHybridDecoder.java                       */
/* ------------------------------------------------------------------------- */

typedef struct {
  int w;
  const uint8_t* in;
  size_t len, pos;
  int packed;         /* mode */
  int32_t count;      /* currentCount (Java int, may go negative: see read_int) */
  int32_t value;      /* currentValue */
  int32_t* buf;       /* currentBuffer */
  int64_t buf_len;    /* currentBuffer.length */
  uint8_t* bytes;
  size_t want;        /* values the caller may still ask for (bounds scratch size only) */
} rle_dec;

/* :80-109 readNext */
static int rle_read_next(rle_dec* d) {
  if (d->pos >= d->len) return ORC_PAST_END; /* :81 */
  uint32_t header;
  int st = orc_read_uvarint32(d->in, d->len, &d->pos, &header);
  if (st) return st;
  d->packed = (int)(header & 1u);
  if (!d->packed) {
    d->count = (int32_t)(header >> 1);
    /* BytesUtils.readIntLittleEndianPaddedOnBitWidth, BytesUtils.java:124-142: not masked to w */
    int nb = (d->w + 7) / 8;
    if (d->len - d->pos < (size_t)nb) {
      d->pos = d->len;
      return ORC_EOF;
    }
    uint32_t x = 0;
    for (int i = 0; i < nb; ++i) x += ((uint32_t)d->in[d->pos + (size_t)i]) << (8 * i);
    d->pos += (size_t)nb;
    d->value = (int32_t)x;
    return ORC_OK;
  }
  uint32_t groups = header >> 1;
  if (groups > 0x0FFFFFFFu) return ORC_DECODE; /* groups*8 overflows a Java int: negative array size */
  d->count = (int32_t)(groups * 8u);
  d->buf_len = d->count;
  /* :94-99: per-run scratch; only the part that can still be asked for is materialised */
  size_t vals = (size_t)d->count;
  size_t need = (d->want + 7) & ~(size_t)7;
  if (vals > need) vals = need;
  free(d->buf);
  free(d->bytes);
  d->buf = (int32_t*)calloc(vals ? vals : 1, sizeof(int32_t));
  d->bytes = (uint8_t*)calloc((vals / 8) * (size_t)d->w + 1, 1);
  if (!d->buf || !d->bytes) return ORC_DECODE;
  uint64_t to_read = (uint64_t)groups * (uint64_t)d->w; /* ceil(count*w/8) == groups*w */
  size_t avail = d->len - d->pos;
  if (to_read > avail) to_read = avail; /* :98 -- short final run, the rest stays zero */
  size_t copy = (size_t)to_read;
  if (copy > (vals / 8) * (size_t)d->w) copy = (vals / 8) * (size_t)d->w;
  memcpy(d->bytes, d->in + d->pos, copy);
  d->pos += (size_t)to_read;
  for (size_t vi = 0, bi = 0; vi < vals; vi += 8, bi += (size_t)d->w)
    orc_unpack8_32(0, d->w, d->bytes + bi, d->buf + vi); /* :100-104 */
  return ORC_OK;
}

/* :61-78 readInt.  A zero-count RLE run leaves currentCount negative, so the
 * run value keeps being returned without another readNext; a zero-group packed
 * run indexes an empty array (ArrayIndexOutOfBounds). */
static int rle_read_int(rle_dec* d, int32_t* out) {
  if (d->count == 0) {
    int st = rle_read_next(d);
    if (st) return st;
  }
  d->count = (int32_t)((uint32_t)d->count - 1u);
  if (!d->packed) {
    *out = d->value;
    return ORC_OK;
  }
  int64_t idx = d->buf_len - 1 - (int64_t)d->count;
  if (idx < 0 || idx >= d->buf_len) return ORC_DECODE;
  *out = d->buf[idx];
  return ORC_OK;
}

int orc_rle_decode(int w, const uint8_t* in, size_t in_len, int32_t* out, size_t n,
                   size_t* written, size_t* consumed) {
  if (written) *written = 0;
  if (consumed) *consumed = 0;
  if (w < 0 || w > 32) return ORC_BAD_ARG; /* :55 */
  rle_dec d;
  memset(&d, 0, sizeof d);
  d.w = w;
  d.in = in;
  d.len = in_len;
  int st = ORC_OK;
  size_t i = 0;
  for (; i < n; ++i) {
    d.want = n - i;
    st = rle_read_int(&d, &out[i]);
    if (st) break;
  }
  free(d.buf);
  free(d.bytes);
  if (written) *written = i;
  if (consumed) *consumed = d.pos;
  return st;
}

/* COL/values/rle/RunLengthBitPackingHybridValuesReader.java:41-45 */
int orc_rle_prefixed_decode(int w, const uint8_t* in, size_t in_len, int32_t* out, size_t n,
                            size_t* written, size_t* consumed) {
  if (written) *written = 0;
  if (consumed) *consumed = 0;
  if (in_len < 4) return ORC_EOF; /* BytesUtils.readIntLittleEndian, BytesUtils.java:85-95 */
  int32_t length = (int32_t)((uint32_t)in[0] | ((uint32_t)in[1] << 8) | ((uint32_t)in[2] << 16) |
                             ((uint32_t)in[3] << 24));
  if (length < 0) return ORC_DECODE;
  if ((size_t)length > in_len - 4) return ORC_EOF; /* sliceStream(length) */
  size_t used;
  int st = orc_rle_decode(w, in + 4, (size_t)length, out, n, written, &used);
  if (consumed) *consumed = 4 + (size_t)length; /* updateNextOffset(length + 4) */
  return st;
}

/* ------------------------------------------------------------------------- */
/* RLE / bit-packing hybrid encoder (used to synthesise pages as parquet-java  */
/* writes them).  COL/values/rle/RunLengthBitPackingHybridEncoder.java         */
/* ------------------------------------------------------------------------- */

typedef struct {
  int w;
  uint8_t* out;
  size_t cap, size;
  int32_t previous;
  int32_t buffered[8];
  int n_buffered;
  int repeat;
  int groups;
  long header_at; /* bitPackedRunHeaderPointer */
  int overflow;
} rle_enc;

static void enc_put(rle_enc* e, const uint8_t* p, size_t n) {
  if (e->size + n > e->cap) {
    e->overflow = 1;
    return;
  }
  memcpy(e->out + e->size, p, n);
  e->size += n;
}

static void enc_end_packed(rle_enc* e) { /* :218-235 */
  if (e->header_at == -1) return;
  if (!e->overflow) e->out[e->header_at] = (uint8_t)((e->groups << 1) | 1);
  e->header_at = -1;
  e->groups = 0;
}

static void enc_write_or_append_packed(rle_enc* e) { /* :185-210 */
  if (e->groups >= 63) enc_end_packed(e);
  if (e->header_at == -1) {
    uint8_t z = 0;
    e->header_at = (long)e->size;
    enc_put(e, &z, 1);
  }
  uint8_t pack[32];
  orc_pack8_32(0, e->w, e->buffered, pack);
  enc_put(e, pack, (size_t)e->w);
  e->n_buffered = 0;
  e->repeat = 0;
  ++e->groups;
}

static void enc_write_rle(rle_enc* e) { /* :237-253 */
  enc_end_packed(e);
  uint8_t tmp[8];
  size_t n = put_uvarint32((uint32_t)e->repeat << 1, tmp);
  enc_put(e, tmp, n);
  int nb = (e->w + 7) / 8; /* BytesUtils.writeIntLittleEndianPaddedOnBitWidth */
  for (int i = 0; i < nb; ++i) tmp[i] = (uint8_t)(((uint32_t)e->previous >> (8 * i)) & 0xFF);
  enc_put(e, tmp, (size_t)nb);
  e->repeat = 0;
  e->n_buffered = 0;
}

static void enc_write_int(rle_enc* e, int32_t v) { /* :146-183 */
  if (v == e->previous) {
    ++e->repeat;
    if (e->repeat >= 8) return;
  } else {
    if (e->repeat >= 8) enc_write_rle(e);
    e->repeat = 1;
    e->previous = v;
  }
  e->buffered[e->n_buffered++] = v;
  if (e->n_buffered == 8) enc_write_or_append_packed(e);
}

long orc_rle_encode(int w, const int32_t* v, size_t n, uint8_t* out, size_t cap) {
  if (w < 0 || w > 32) return -1;
  rle_enc e;
  memset(&e, 0, sizeof e);
  e.w = w;
  e.out = out;
  e.cap = cap;
  e.header_at = -1;
  for (size_t i = 0; i < n; ++i) enc_write_int(&e, v[i]);
  /* toBytes, :255-273 */
  if (e.repeat >= 8) {
    enc_write_rle(&e);
  } else if (e.n_buffered > 0) {
    for (int i = e.n_buffered; i < 8; ++i) e.buffered[i] = 0;
    enc_write_or_append_packed(&e);
    enc_end_packed(&e);
  } else {
    enc_end_packed(&e);
  }
  return e.overflow ? -1 : (long)e.size;
}

/* ------------------------------------------------------------------------- */
/* Dictionary-encoded data pages                                              */
/* COL/values/dictionary/DictionaryValuesReader.java:49-127                   */
/* ------------------------------------------------------------------------- */

int orc_dict_decode_ids(const uint8_t* page, size_t len, int32_t* ids, size_t n, size_t* written) {
  if (written) *written = 0;
  if (len == 0) return n ? ORC_DECODE : ORC_OK; /* :56-63 "Attempt to read from empty page" */
  int w = page[0];                             /* :53 */
  return orc_rle_decode(w, page + 1, len - 1, ids, n, written, NULL);
}

int orc_dict_decode(const uint8_t* page, size_t len, const void* dict, size_t dict_n,
                    int elem_size, void* out, size_t n, size_t* written) {
  if (written) *written = 0;
  if (len == 0) return n ? ORC_DECODE : ORC_OK;
  int w = page[0];
  if (w > 32) return ORC_BAD_ARG;
  rle_dec d;
  memset(&d, 0, sizeof d);
  d.w = w;
  d.in = page + 1;
  d.len = len - 1;
  int st = ORC_OK;
  size_t i = 0;
  for (; i < n; ++i) {
    int32_t id;
    d.want = n - i;
    st = rle_read_int(&d, &id);
    if (st) break;
    /* Dictionary.decodeToX(id): plain Java array load (COL/values/dictionary/PlainValuesDictionary.java:148-345) */
    if (id < 0 || (size_t)id >= dict_n) {
      st = ORC_DICT_OOB;
      break;
    }
    memcpy((uint8_t*)out + i * (size_t)elem_size, (const uint8_t*)dict + (size_t)id * (size_t)elem_size,
           (size_t)elem_size);
  }
  free(d.buf);
  free(d.bytes);
  if (written) *written = i;
  return st;
}

/* DictionaryValuesWriter.getBytes, COL/values/dictionary/DictionaryValuesWriter.java:153-180 */
long orc_dict_encode_ids(const int32_t* ids, size_t n, size_t dict_n, uint8_t* out, size_t cap) {
  if (cap < 1) return -1;
  int32_t max_id = (int32_t)dict_n - 1;
  int w = 0; /* BytesUtils.getWidthFromMaxInt: 32 - numberOfLeadingZeros(bound) */
  for (uint32_t b = (uint32_t)max_id; b; b >>= 1) ++w;
  out[0] = (uint8_t)w;
  long body = orc_rle_encode(w, ids, n, out + 1, cap - 1);
  return body < 0 ? -1 : body + 1;
}

/* ------------------------------------------------------------------------- */
/* DELTA_BINARY_PACKED                                                        */
/* COL/values/delta/DeltaBinaryPackingValuesReader.java                       */
/* ------------------------------------------------------------------------- */

int orc_delta_decode(const uint8_t* in, size_t in_len, int64_t* out, size_t cap, size_t* total,
                     size_t* consumed) {
  size_t pos = 0;
  if (total) *total = 0;
  if (consumed) *consumed = 0;
  uint32_t block, mini_n, count_u;
  int st;
  /* DeltaBinaryPackingConfig.readConfig, DeltaBinaryPackingConfig.java:35-45 */
  if ((st = orc_read_uvarint32(in, in_len, &pos, &block))) return st;
  if ((st = orc_read_uvarint32(in, in_len, &pos, &mini_n))) return st;
  int32_t block_i = (int32_t)block, mini_i = (int32_t)mini_n;
  if (block_i < 0 || mini_i < 0) return ORC_DECODE; /* sizes beyond a Java int: nothing sane follows */
  if (mini_i == 0) return ORC_BAD_ARG;           /* x/0.0 % 8 is NaN: precondition fails */
  if (block_i % mini_i != 0) return ORC_BAD_ARG; /* fractional mini size is no multiple of 8 */
  int32_t mini_size = block_i / mini_i;
  if (mini_size % 8 != 0) return ORC_BAD_ARG;
  if ((st = orc_read_uvarint32(in, in_len, &pos, &count_u))) return st; /* :64 */
  int32_t count = (int32_t)count_u;
  if (count < 0) return ORC_DECODE; /* negative array size */
  if (total) *total = (size_t)count;
  if ((size_t)count > cap) return ORC_DECODE;
  int64_t first;
  if ((st = orc_read_zigzag_varlong(in, in_len, &pos, &first))) return st; /* :69 */
  if (count > 1 && (mini_size <= 0 || mini_i < 0)) return ORC_DECODE;      /* the Java loop would not advance */
  /* :83-87 allocateValuesBuffer: whole miniblocks + 1 */
  size_t buf_n = 1;
  if (mini_size > 0) buf_n = ((size_t)count + (size_t)mini_size - 1) / (size_t)mini_size * (size_t)mini_size + 1;
  int64_t* buf = (int64_t*)calloc(buf_n + 8, sizeof(int64_t));
  int* widths = (int*)calloc((size_t)(mini_i > 0 ? mini_i : 1), sizeof(int));
  if (!buf || !widths) {
    free(buf);
    free(widths);
    return ORC_DECODE;
  }
  size_t buffered = 0;
  buf[buffered++] = first;
  st = ORC_OK;
  while (buffered < (size_t)count) { /* :71-75 */
    /* loadNewBlockToBuffer, :121-143 */
    int64_t min_delta;
    if (orc_read_zigzag_varlong(in, in_len, &pos, &min_delta)) {
      st = ORC_DECODE; /* "can not read min delta in current block" */
      break;
    }
    int bad = 0;
    for (int i = 0; i < mini_i; ++i) { /* :164-172 */
      if (pos >= in_len) {
        bad = 1;
        break;
      }
      widths[i] = in[pos++];
    }
    if (bad) {
      st = ORC_DECODE; /* "Can not decode bitwidth in block header" */
      break;
    }
    int i;
    for (i = 0; i < mini_i && buffered < (size_t)count; ++i) {
      int w = widths[i];
      if (w > 64) { /* Packer.newBytePackerForLong: packer table has 65 entries */
        st = ORC_DECODE;
        break;
      }
      for (int j = 0; j < mini_size; j += 8) { /* :150-162 */
        if (in_len - pos < (size_t)w) {
          st = ORC_EOF; /* in.slice(w) */
          break;
        }
        if (w == 0)
          memset(buf + buffered, 0, 8 * sizeof(int64_t));
        else
          orc_unpack8_64(0, w, in + pos, buf + buffered);
        pos += (size_t)w;
        buffered += 8;
      }
      if (st) break;
    }
    if (st) break;
    size_t unpacked = (size_t)i * (size_t)mini_size;
    for (size_t j = buffered - unpacked; j < buffered; ++j) /* :137-142, wrapping adds */
      buf[j] = (int64_t)((uint64_t)buf[j] + (uint64_t)min_delta + (uint64_t)buf[j - 1]);
  }
  if (!st) memcpy(out, buf, (size_t)count * sizeof(int64_t));
  free(buf);
  free(widths);
  if (consumed) *consumed = pos;
  return st;
}

/* COL/values/delta/DeltaBinaryPackingValuesWriterForLong.java:75-184.  The block
 * buffer and the width array are reused between flushes, so a short last block
 * packs stale slots and repeats stale widths exactly as the Java does. */
long orc_delta_encode_i64(const int64_t* v, size_t n, int block, int mini, uint8_t* out, size_t cap) {
  if (mini <= 0 || block <= 0 || block % mini || (block / mini) % 8) return -1;
  int mini_size = block / mini;
  int64_t* deltas = (int64_t*)calloc((size_t)block, sizeof(int64_t));
  int* widths = (int*)calloc((size_t)mini, sizeof(int));
  uint8_t* body = (uint8_t*)malloc(n * 9 + (size_t)block * 8 + 64 + (n / (size_t)block + 2) * (size_t)(10 + mini));
  if (!deltas || !widths || !body) return -1;
  size_t bs = 0;
  int64_t first = 0, prev = 0, min_delta = INT64_MAX;
  int to_flush = 0;
  for (size_t idx = 0; idx <= n; ++idx) {
    int flush = 0;
    if (idx < n) {
      if (idx == 0) {
        first = prev = v[0];
        continue;
      }
      int64_t delta = (int64_t)((uint64_t)v[idx] - (uint64_t)prev);
      prev = v[idx];
      deltas[to_flush++] = delta;
      if (delta < min_delta) min_delta = delta;
      flush = (to_flush == block);
    } else {
      flush = (to_flush != 0); /* getBytes */
    }
    if (!flush) continue;
    for (int i = 0; i < to_flush; ++i) deltas[i] = (int64_t)((uint64_t)deltas[i] - (uint64_t)min_delta);
    bs += put_uvarlong(((uint64_t)min_delta << 1) ^ (uint64_t)(min_delta >> 63), body + bs);
    int minis = (to_flush + mini_size - 1) / mini_size;
    for (int m = 0; m < minis; ++m) {
      uint64_t mask = 0;
      int end = (m + 1) * mini_size < to_flush ? (m + 1) * mini_size : to_flush;
      for (int i = m * mini_size; i < end; ++i) mask |= (uint64_t)deltas[i];
      int w = 0;
      while (mask) {
        ++w;
        mask >>= 1;
      }
      widths[m] = w;
    }
    for (int m = 0; m < mini; ++m) body[bs++] = (uint8_t)widths[m];
    for (int m = 0; m < minis; ++m) {
      for (int j = m * mini_size; j < (m + 1) * mini_size; j += 8) {
        orc_pack8_64(0, widths[m], deltas + j, body + bs);
        bs += (size_t)widths[m];
      }
    }
    min_delta = INT64_MAX;
    to_flush = 0;
  }
  uint8_t hdr[40];
  size_t hs = 0;
  hs += put_uvarint32((uint32_t)block, hdr + hs);
  hs += put_uvarint32((uint32_t)mini, hdr + hs);
  hs += put_uvarint32((uint32_t)n, hdr + hs);
  hs += put_uvarlong(((uint64_t)first << 1) ^ (uint64_t)(first >> 63), hdr + hs);
  long ret = -1;
  if (hs + bs <= cap) {
    memcpy(out, hdr, hs);
    memcpy(out + hs, body, bs);
    ret = (long)(hs + bs);
  }
  free(deltas);
  free(widths);
  free(body);
  return ret;
}

/* COL/values/delta/DeltaBinaryPackingValuesWriterForInteger.java:75-182: 32-bit
 * wrapping deltas, 32-bit zig-zag varints, int packer (widths 0..32). */
long orc_delta_encode_i32(const int32_t* v, size_t n, int block, int mini, uint8_t* out, size_t cap) {
  if (mini <= 0 || block <= 0 || block % mini || (block / mini) % 8) return -1;
  int mini_size = block / mini;
  int32_t* deltas = (int32_t*)calloc((size_t)block, sizeof(int32_t));
  int* widths = (int*)calloc((size_t)mini, sizeof(int));
  uint8_t* body = (uint8_t*)malloc(n * 5 + (size_t)block * 4 + 64 + (n / (size_t)block + 2) * (size_t)(5 + mini));
  if (!deltas || !widths || !body) return -1;
  size_t bs = 0;
  int32_t first = 0, prev = 0, min_delta = INT32_MAX;
  int to_flush = 0;
  for (size_t idx = 0; idx <= n; ++idx) {
    int flush = 0;
    if (idx < n) {
      if (idx == 0) {
        first = prev = v[0];
        continue;
      }
      int32_t delta = (int32_t)((uint32_t)v[idx] - (uint32_t)prev);
      prev = v[idx];
      deltas[to_flush++] = delta;
      if (delta < min_delta) min_delta = delta;
      flush = (to_flush == block);
    } else {
      flush = (to_flush != 0);
    }
    if (!flush) continue;
    for (int i = 0; i < to_flush; ++i) deltas[i] = (int32_t)((uint32_t)deltas[i] - (uint32_t)min_delta);
    bs += put_uvarint32(((uint32_t)min_delta << 1) ^ (uint32_t)(min_delta >> 31), body + bs);
    int minis = (to_flush + mini_size - 1) / mini_size;
    for (int m = 0; m < minis; ++m) {
      uint32_t mask = 0;
      int end = (m + 1) * mini_size < to_flush ? (m + 1) * mini_size : to_flush;
      for (int i = m * mini_size; i < end; ++i) mask |= (uint32_t)deltas[i];
      int w = 0;
      while (mask) {
        ++w;
        mask >>= 1;
      }
      widths[m] = w;
    }
    for (int m = 0; m < mini; ++m) body[bs++] = (uint8_t)widths[m];
    for (int m = 0; m < minis; ++m) {
      for (int j = m * mini_size; j < (m + 1) * mini_size; j += 8) {
        orc_pack8_32(0, widths[m], deltas + j, body + bs);
        bs += (size_t)widths[m];
      }
    }
    min_delta = INT32_MAX;
    to_flush = 0;
  }
  uint8_t hdr[40];
  size_t hs = 0;
  hs += put_uvarint32((uint32_t)block, hdr + hs);
  hs += put_uvarint32((uint32_t)mini, hdr + hs);
  hs += put_uvarint32((uint32_t)n, hdr + hs);
  hs += put_uvarint32(((uint32_t)first << 1) ^ (uint32_t)(first >> 31), hdr + hs);
  long ret = -1;
  if (hs + bs <= cap) {
    memcpy(out, hdr, hs);
    memcpy(out + hs, body, bs);
    ret = (long)(hs + bs);
  }
  free(deltas);
  free(widths);
  free(body);
  return ret;
}

/* ------------------------------------------------------------------------- */
/* BYTE_STREAM_SPLIT                                                          */
/* COL/values/bytestreamsplit/ByteStreamSplitValuesReader.java                */
/* ------------------------------------------------------------------------- */

int orc_bss_decode(const uint8_t* in, size_t in_len, int k, size_t value_count_bound, uint8_t* out,
                   size_t* n_values) {
  if (n_values) *n_values = 0;
  if (k <= 0) return ORC_BAD_ARG;
  if (in_len % (size_t)k != 0) return ORC_DECODE; /* :195-200 */
  size_t n = in_len / (size_t)k;
  if (value_count_bound < n) return ORC_DECODE; /* :204-209 */
  for (int s = 0; s < k; ++s) {                 /* :176-183 (the unrolled k=2,4,8,12,16 loops compute the same) */
    const uint8_t* src = in + (size_t)s * n;
    for (size_t i = 0; i < n; ++i) out[i * (size_t)k + (size_t)s] = src[i];
  }
  if (n_values) *n_values = n;
  return ORC_OK;
}

/* ByteStreamSplitValuesWriter scatter (COL/values/bytestreamsplit/ByteStreamSplitValuesWriter.java) */
void orc_bss_encode(const uint8_t* in, size_t n, int k, uint8_t* out) {
  for (size_t i = 0; i < n; ++i)
    for (int s = 0; s < k; ++s) out[(size_t)s * n + i] = in[i * (size_t)k + (size_t)s];
}

/* ------------------------------------------------------------------------- */
/* Legacy BIT_PACKED levels                                                   */
/* COL/values/bitpacking/ByteBitPackingValuesReader.java:41-88 (Packer.BIG_ENDIAN) */
/* ------------------------------------------------------------------------- */

int orc_bitpacked_levels_decode(int w, const uint8_t* in, size_t in_len, int32_t* out, size_t n,
                                size_t* consumed) {
  if (consumed) *consumed = 0;
  if (w < 0 || w > 32) return ORC_BAD_ARG;
  size_t length = (n * (size_t)w + 7) / 8; /* :78-79 */
  if (length > in_len) length = in_len;    /* :84 */
  size_t pos = 0;
  int32_t decoded[8];
  uint8_t temp[32];
  for (size_t i = 0; i < n; ++i) {
    if ((i & 7) == 0) { /* readMore, :48-65 */
      size_t avail = length - pos;
      size_t take = avail < (size_t)w ? avail : (size_t)w;
      memset(temp, 0, sizeof temp);
      memcpy(temp, in + pos, take);
      pos += take;
      memset(decoded, 0, sizeof decoded);
      orc_unpack8_32(1, w, temp, decoded);
    }
    out[i] = decoded[i & 7];
  }
  if (consumed) *consumed = length; /* updateNextOffset(length) */
  return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* BINARY / FIXED_LEN_BYTE_ARRAY / INT96 dictionaries                          */
/* COL/values/dictionary/PlainValuesDictionary.java:88-115 (PlainBinaryDictionary), selected by        */
/* COL/Encoding.java:89-112 (BINARY: length-prefixed; FLBA: typeLength; INT96: 12)                     */
/* ------------------------------------------------------------------------- */

/* A view = what Binary.fromConstantByteBuffer(dictionaryBytes, offset, len) holds: the low word is
 * the offset of the first byte in the dictionary page, the high word the length.  The Java object is
 * created without looking at the bytes, so an entry that runs past the page (or has a negative
 * length) only fails when it is read: such entries get ORC_VIEW_INVALID.  The 4-byte length prefix
 * itself is read eagerly (:97), outside the page that is an IndexOutOfBoundsException -> ORC_DECODE. */
int orc_binary_dict_views(const uint8_t* page, size_t len, size_t n, int fixed_length, uint64_t* views) {
  int64_t offset = 0;
  if (fixed_length < 0) return ORC_BAD_ARG; /* Preconditions.checkArgument(length > 0), :107 */
  for (size_t i = 0; i < n; ++i) {
    int32_t l;
    if (fixed_length == 0) {
      if (offset < 0 || (uint64_t)offset + 4 > len) return ORC_DECODE; /* readIntLittleEndian, :97 */
      l = (int32_t)((uint32_t)page[offset] | ((uint32_t)page[offset + 1] << 8) |
                    ((uint32_t)page[offset + 2] << 16) | ((uint32_t)page[offset + 3] << 24));
      offset += 4; /* :99 */
    } else {
      l = fixed_length;
    }
    if (l < 0 || offset < 0 || (uint64_t)offset + (uint64_t)l > len || offset > 0x7fffffff)
      views[i] = ORC_VIEW_INVALID;
    else
      views[i] = ((uint64_t)(uint32_t)l << 32) | (uint32_t)offset;
    offset += l; /* :103, :112 (Java int arithmetic; a negative length walks backwards) */
    if (offset > 0x7fffffffLL || offset < -0x80000000LL) return ORC_DECODE; /* would wrap a Java int: treated as malformed */
  }
  return ORC_OK;
}

/* The read loop `for i: Binary b = reader.readBytes(); out.write(b.getBytes())` over a dictionary
 * page whose ids were already turned into views (DictionaryValuesReader.readBytes,
 * COL/values/dictionary/DictionaryValuesReader.java:76-83 -> Dictionary.decodeToBinary), gathered
 * into one buffer: offsets[i] = position of value i, offsets[written] = end of the last one. */
int orc_binary_materialize(const uint64_t* views, size_t n, const uint8_t* dict_bytes, size_t dict_len,
                           int32_t* offsets, uint8_t* bytes, size_t cap, size_t* written, size_t* bytes_written) {
  size_t pos = 0, i = 0;
  int st = ORC_OK;
  if (cap > 0x7fffffff) cap = 0x7fffffff; /* int32 offsets */
  for (; i < n; ++i) {
    uint64_t v = views[i];
    uint32_t off = (uint32_t)v, l = (uint32_t)(v >> 32);
    if (l == 0xffffffffu || (uint64_t)off + l > dict_len) {
      st = ORC_DECODE;
      break;
    }
    if (pos + l > cap) {
      st = ORC_BAD_ARG;
      break;
    }
    offsets[i] = (int32_t)pos;
    memcpy(bytes + pos, dict_bytes + off, l);
    pos += l;
  }
  offsets[i] = (int32_t)pos;
  if (written) *written = i;
  if (bytes_written) *bytes_written = pos;
  return st;
}
