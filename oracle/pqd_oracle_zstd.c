/* CPU oracle, page decompression: Zstandard frames.  TEST INFRASTRUCTURE ONLY (see pqd_oracle.h).
 *
 * The reference hands ZSTD page bodies to com.github.luben:zstd-jni 1.5.7-13 (ZstdDecompressorStream.java:
 * 30-60 wraps ZstdInputStreamNoFinalizer; pom.xml:114), a binding of the C library; neither is in the
 * reference tree.  This file restates the published format (RFC 8878): frame header, raw / RLE /
 * compressed blocks, literals (raw, RLE, Huffman-coded in one or four streams, tree given directly, by
 * FSE-coded weights or reused), sequences (FSE-coded literal-length / offset / match-length codes with
 * predefined, RLE, described or repeated tables, the three repeat offsets), optional content checksum
 * (XXH64).  No dictionaries (parquet pages carry none).  Pinned by streams the C library produced
 * (tests/golden/codec_vectors.json at two compression levels, ZSTD pages written by pyarrow).
 * As for the other codecs, the page must produce exactly `cap` bytes. */
#include <stdlib.h>
#include <string.h>
#include "pqd_oracle.h"

#define ZERR ORC_DECODE

/* ---- bit readers ---- */
typedef struct {
  const uint8_t* p;
  size_t n;
  size_t bit; /* forward: bits consumed */
} FwdBits;

static uint32_t fwd_peek(const FwdBits* b, int k) { /* k <= 25; bits beyond the buffer read as zero */
  uint64_t v = 0;
  size_t byte = b->bit >> 3;
  for (int i = 0; i < 5; ++i)
    if (byte + i < b->n) v |= (uint64_t)b->p[byte + i] << (8 * i);
  return (uint32_t)((v >> (b->bit & 7)) & ((1u << k) - 1u));
}

typedef struct {
  const uint8_t* p;
  long long bit; /* bits not yet read, counted from the stream's first byte; negative once over-read */
} BackBits;

static int back_init(BackBits* b, const uint8_t* p, size_t n) {
  if (n == 0 || p[n - 1] == 0) return ZERR;
  int hb = 7;
  while (!(p[n - 1] >> hb)) --hb;
  b->p = p;
  b->bit = (long long)(n - 1) * 8 + hb;
  return ORC_OK;
}

static uint64_t back_read(BackBits* b, int k) { /* k <= 32; bits in front of the stream read as zero */
  if (k == 0) return 0;
  b->bit -= k;
  long long lo = b->bit;
  uint64_t v = 0;
  for (int i = 0; i < k; ++i) {
    long long pos = lo + i;
    if (pos >= 0) v |= (uint64_t)((b->p[pos >> 3] >> (pos & 7)) & 1u) << i;
  }
  return v;
}

/* ---- FSE ---- */
typedef struct {
  uint8_t symbol, nbits;
  uint16_t base;
} FseEntry;

typedef struct {
  FseEntry e[512];
  int log;
} FseTable;

static int highbit(uint32_t v) {
  int r = 0;
  while (v >>= 1) ++r;
  return r;
}

static int fse_build(FseTable* t, const int16_t* norm, int n_sym, int log) {
  const int size = 1 << log;
  uint16_t next[256];
  int high = size - 1;
  t->log = log;
  for (int s = 0; s < n_sym; ++s) {
    if (norm[s] == -1) {
      t->e[high--].symbol = (uint8_t)s;
      next[s] = 1;
    } else {
      next[s] = (uint16_t)norm[s];
    }
  }
  const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
  int pos = 0;
  for (int s = 0; s < n_sym; ++s) {
    for (int i = 0; i < norm[s]; ++i) {
      t->e[pos].symbol = (uint8_t)s;
      do pos = (pos + step) & mask;
      while (pos > high);
    }
  }
  if (pos != 0) return ZERR;
  for (int u = 0; u < size; ++u) {
    const int s = t->e[u].symbol;
    const uint32_t ns = next[s]++;
    const int nb = log - highbit(ns);
    t->e[u].nbits = (uint8_t)nb;
    t->e[u].base = (uint16_t)((ns << nb) - size);
  }
  return ORC_OK;
}

/* normalized counts (RFC 8878 4.1.1); returns bytes used or -1 */
static long fse_read_norm(const uint8_t* p, size_t n, int16_t* norm, int* n_sym, int max_sym, int* log, int max_log) {
  FwdBits b = {p, n, 0};
  const int al = 5 + (int)fwd_peek(&b, 4);
  b.bit += 4;
  if (al > max_log) return -1;
  int remaining = (1 << al) + 1, threshold = 1 << al, nbits = al + 1, sym = 0, prev0 = 0;
  while (remaining > 1 && sym <= max_sym) {
    if (prev0) {
      for (;;) {
        const int rep = (int)fwd_peek(&b, 2);
        b.bit += 2;
        for (int i = 0; i < rep && sym <= max_sym; ++i) norm[sym++] = 0;
        if (rep != 3) break;
      }
      if (sym > max_sym) break;
    }
    const int max = (2 * threshold - 1) - remaining;
    const uint32_t bits = fwd_peek(&b, nbits);
    int count;
    if ((int)(bits & (uint32_t)(threshold - 1)) < max) {
      count = (int)(bits & (uint32_t)(threshold - 1));
      b.bit += (size_t)(nbits - 1);
    } else {
      count = (int)(bits & (uint32_t)(2 * threshold - 1));
      if (count >= threshold) count -= max;
      b.bit += (size_t)nbits;
    }
    --count; /* 0 stands for "less than one": -1 */
    remaining -= count < 0 ? -count : count;
    norm[sym++] = (int16_t)count;
    prev0 = count == 0;
    while (remaining < threshold) {
      --nbits;
      threshold >>= 1;
    }
    if (b.bit > n * 8 + 32) return -1;
  }
  if (remaining != 1 || sym > max_sym + 1) return -1;
  *n_sym = sym;
  *log = al;
  const size_t used = (b.bit + 7) >> 3;
  return used > n ? -1 : (long)used;
}

/* ---- Huffman ---- */
typedef struct {
  uint8_t symbol[2048], nbits[2048];
  int max_bits;
  int valid;
} HufTable;

static int huf_build(HufTable* h, uint8_t* w, int n_given) {
  /* the last weight completes the sum of 2^(w-1) to a power of two */
  uint32_t total = 0;
  for (int i = 0; i < n_given; ++i) {
    if (w[i] > 11) return ZERR;
    total += w[i] ? 1u << (w[i] - 1) : 0u;
  }
  if (total == 0) return ZERR;
  const int max_bits = highbit(total) + 1;
  if (max_bits > 11) return ZERR;
  const uint32_t rest = (1u << max_bits) - total;
  if (rest & (rest - 1)) return ZERR;
  w[n_given] = (uint8_t)(highbit(rest) + 1);
  const int n = n_given + 1;
  uint32_t start[13] = {0}, cnt[13] = {0};
  for (int i = 0; i < n; ++i) ++cnt[w[i]];
  uint32_t at = 0;
  for (int r = 1; r <= max_bits; ++r) {
    start[r] = at;
    at += cnt[r] << (r - 1);
  }
  if (at != (1u << max_bits)) return ZERR;
  for (int s = 0; s < n; ++s) {
    if (!w[s]) continue;
    const uint32_t len = 1u << (w[s] - 1);
    for (uint32_t i = 0; i < len; ++i) {
      h->symbol[start[w[s]] + i] = (uint8_t)s;
      h->nbits[start[w[s]] + i] = (uint8_t)(max_bits + 1 - w[s]);
    }
    start[w[s]] += len;
  }
  h->max_bits = max_bits;
  h->valid = 1;
  return ORC_OK;
}

/* tree description; returns bytes used or -1 */
static long huf_read_tree(HufTable* h, const uint8_t* p, size_t n) {
  if (n == 0) return -1;
  uint8_t w[256];
  int n_w = 0;
  size_t used;
  const int hb = p[0];
  if (hb >= 128) { /* direct: 4 bits each */
    n_w = hb - 127;
    used = 1 + (size_t)(n_w + 1) / 2;
    if (used > n) return -1;
    for (int i = 0; i < n_w; ++i) w[i] = (i & 1) ? p[1 + i / 2] & 15 : p[1 + i / 2] >> 4;
  } else { /* FSE-coded weights, two interleaved states */
    used = 1 + (size_t)hb;
    if (used > n || hb == 0) return -1;
    int16_t norm[256];
    int n_sym, log;
    const long tb = fse_read_norm(p + 1, (size_t)hb, norm, &n_sym, 255, &log, 6);
    if (tb < 0) return -1;
    FseTable t;
    if (fse_build(&t, norm, n_sym, log)) return -1;
    BackBits b;
    if (back_init(&b, p + 1 + tb, (size_t)hb - (size_t)tb)) return -1;
    uint32_t s1 = (uint32_t)back_read(&b, log), s2 = (uint32_t)back_read(&b, log);
    for (;;) {
      if (n_w >= 254) return -1;
      w[n_w++] = t.e[s1].symbol;
      if (b.bit < t.e[s1].nbits) { /* the other state still holds one symbol */
        if (n_w >= 255) return -1;
        w[n_w++] = t.e[s2].symbol;
        break;
      }
      s1 = t.e[s1].base + (uint32_t)back_read(&b, t.e[s1].nbits);
      w[n_w++] = t.e[s2].symbol;
      if (b.bit < t.e[s2].nbits) {
        if (n_w >= 255) return -1;
        w[n_w++] = t.e[s1].symbol;
        break;
      }
      s2 = t.e[s2].base + (uint32_t)back_read(&b, t.e[s2].nbits);
    }
  }
  if (huf_build(h, w, n_w)) return -1;
  return (long)used;
}

static int huf_stream(const HufTable* h, const uint8_t* p, size_t n, uint8_t* out, size_t count) {
  BackBits b;
  if (back_init(&b, p, n)) return ZERR;
  for (size_t i = 0; i < count; ++i) {
    /* peek max_bits: read, then give back what the code does not use */
    const uint32_t v = (uint32_t)back_read(&b, h->max_bits);
    out[i] = h->symbol[v];
    b.bit += h->max_bits - h->nbits[v];
  }
  return b.bit == 0 ? ORC_OK : ZERR;
}

/* ---- sequences ---- */
static const int16_t LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                       2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int16_t OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
static const uint32_t LL_BASE[36] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,   10,  11,  12,   13,   14,   15,   16,    18,
                                     20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
static const uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint32_t ML_BASE[53] = {3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                     21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41,
                                     43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
static const uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                    0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

typedef struct {
  FseTable ll, of, ml;
  int have_ll, have_of, have_ml;
  HufTable huf;
  uint32_t rep[3];
} FrameState;

/* one table of the sequences section; returns bytes used or -1 */
static long seq_table(FseTable* t, int* have, int mode, const uint8_t* p, size_t n, const int16_t* def, int def_n, int def_log,
                      int max_sym, int max_log) {
  if (mode == 0) {
    if (fse_build(t, def, def_n, def_log)) return -1;
    *have = 1;
    return 0;
  }
  if (mode == 1) { /* one symbol, no bits */
    if (n < 1 || p[0] > max_sym) return -1;
    t->log = 0;
    t->e[0].symbol = p[0];
    t->e[0].nbits = 0;
    t->e[0].base = 0;
    *have = 1;
    return 1;
  }
  if (mode == 2) {
    int16_t norm[64];
    int n_sym, log;
    const long used = fse_read_norm(p, n, norm, &n_sym, max_sym, &log, max_log);
    if (used < 0 || fse_build(t, norm, n_sym, log)) return -1;
    *have = 1;
    return used;
  }
  return *have ? 0 : -1; /* repeat */
}

static int copy_match(uint8_t* out, size_t* op, size_t cap, size_t offset, size_t len) {
  if (offset == 0 || offset > *op || len > cap - *op) return ZERR;
  for (size_t i = 0; i < len; ++i) out[*op + i] = out[*op - offset + i];
  *op += len;
  return ORC_OK;
}

static int compressed_block(FrameState* fs, const uint8_t* p, size_t n, uint8_t* out, size_t* op, size_t cap) {
  if (n < 1) return ZERR;
  /* ---- literals */
  const int lt = p[0] & 3, sf = (p[0] >> 2) & 3;
  size_t hdr, regen, comp = 0;
  int streams = 1;
  if (lt < 2) {
    if (sf == 0 || sf == 2) {
      hdr = 1;
      regen = p[0] >> 3;
    } else if (sf == 1) {
      hdr = 2;
      if (n < 2) return ZERR;
      regen = (p[0] >> 4) | ((size_t)p[1] << 4);
    } else {
      hdr = 3;
      if (n < 3) return ZERR;
      regen = (p[0] >> 4) | ((size_t)p[1] << 4) | ((size_t)p[2] << 12);
    }
  } else {
    const int bits = sf < 2 ? 10 : sf == 2 ? 14 : 18;
    hdr = sf < 2 ? 3 : sf == 2 ? 4 : 5;
    streams = sf == 0 ? 1 : 4;
    if (n < hdr) return ZERR;
    uint64_t h = 0;
    for (size_t i = 0; i < hdr; ++i) h |= (uint64_t)p[i] << (8 * i);
    regen = (size_t)((h >> 4) & ((1u << bits) - 1u));
    comp = (size_t)((h >> (4 + bits)) & ((1u << bits) - 1u));
  }
  if (regen > 128 * 1024) return ZERR;
  uint8_t* lit = (uint8_t*)malloc(regen + 8);
  if (!lit) return ZERR;
  size_t pos = hdr;
  int st = ORC_OK;
  if (lt == 0) {
    if (n - pos < regen) st = ZERR;
    else memcpy(lit, p + pos, regen);
    pos += regen;
  } else if (lt == 1) {
    if (n - pos < 1) st = ZERR;
    else memset(lit, p[pos], regen);
    pos += 1;
  } else {
    if (n - pos < comp) st = ZERR;
    const uint8_t* q = p + pos;
    size_t left = comp;
    if (!st && lt == 2) {
      const long used = huf_read_tree(&fs->huf, q, left);
      if (used < 0) st = ZERR;
      else {
        q += used;
        left -= (size_t)used;
      }
    } else if (!st && !fs->huf.valid) {
      st = ZERR;
    }
    if (!st && streams == 1) {
      st = huf_stream(&fs->huf, q, left, lit, regen);
    } else if (!st) {
      if (left < 6) st = ZERR;
      else {
        const size_t s1 = q[0] | ((size_t)q[1] << 8), s2 = q[2] | ((size_t)q[3] << 8), s3 = q[4] | ((size_t)q[5] << 8);
        const size_t each = (regen + 3) / 4;
        if (6 + s1 + s2 + s3 > left || each * 3 > regen) st = ZERR;
        else {
          const size_t s4 = left - 6 - s1 - s2 - s3;
          st = huf_stream(&fs->huf, q + 6, s1, lit, each);
          if (!st) st = huf_stream(&fs->huf, q + 6 + s1, s2, lit + each, each);
          if (!st) st = huf_stream(&fs->huf, q + 6 + s1 + s2, s3, lit + 2 * each, each);
          if (!st) st = huf_stream(&fs->huf, q + 6 + s1 + s2 + s3, s4, lit + 3 * each, regen - 3 * each);
        }
      }
    }
    pos += comp;
  }
  if (st) {
    free(lit);
    return st;
  }
  /* ---- sequences */
  size_t lit_at = 0;
  if (n - pos < 1) {
    free(lit);
    return ZERR;
  }
  size_t n_seq = p[pos++];
  if (n_seq >= 128) {
    if (n_seq == 255) {
      if (n - pos < 2) st = ZERR;
      else {
        n_seq = (size_t)p[pos] + ((size_t)p[pos + 1] << 8) + 0x7F00;
        pos += 2;
      }
    } else {
      if (n - pos < 1) st = ZERR;
      else {
        n_seq = ((n_seq - 128) << 8) + p[pos];
        pos += 1;
      }
    }
  }
  if (!st && n_seq) {
    if (n - pos < 1) st = ZERR;
    else {
      const int modes = p[pos++];
      if (modes & 3) st = ZERR;
      long used = st ? -1 : seq_table(&fs->ll, &fs->have_ll, modes >> 6, p + pos, n - pos, LL_DEFAULT, 36, 6, 35, 9);
      if (used < 0) st = ZERR;
      else pos += (size_t)used;
      used = st ? -1 : seq_table(&fs->of, &fs->have_of, (modes >> 4) & 3, p + pos, n - pos, OF_DEFAULT, 29, 5, 31, 8);
      if (used < 0) st = ZERR;
      else pos += (size_t)used;
      used = st ? -1 : seq_table(&fs->ml, &fs->have_ml, (modes >> 2) & 3, p + pos, n - pos, ML_DEFAULT, 53, 6, 52, 9);
      if (used < 0) st = ZERR;
      else pos += (size_t)used;
      BackBits b;
      if (!st && back_init(&b, p + pos, n - pos)) st = ZERR;
      if (!st) {
        uint32_t sl = (uint32_t)back_read(&b, fs->ll.log), so = (uint32_t)back_read(&b, fs->of.log), sm = (uint32_t)back_read(&b, fs->ml.log);
        for (size_t i = 0; i < n_seq && !st; ++i) {
          const int oc = fs->of.e[so].symbol, mc = fs->ml.e[sm].symbol, lc = fs->ll.e[sl].symbol;
          if (oc > 31 || mc > 52 || lc > 35) {
            st = ZERR;
            break;
          }
          const uint64_t ov = ((uint64_t)1 << oc) + back_read(&b, oc);
          const size_t ml = ML_BASE[mc] + (size_t)back_read(&b, ML_BITS[mc]);
          const size_t ll = LL_BASE[lc] + (size_t)back_read(&b, LL_BITS[lc]);
          if (i + 1 < n_seq) {
            sl = fs->ll.e[sl].base + (uint32_t)back_read(&b, fs->ll.e[sl].nbits);
            sm = fs->ml.e[sm].base + (uint32_t)back_read(&b, fs->ml.e[sm].nbits);
            so = fs->of.e[so].base + (uint32_t)back_read(&b, fs->of.e[so].nbits);
          }
          if (b.bit < 0) {
            st = ZERR;
            break;
          }
          size_t offset;
          if (ov > 3) {
            offset = (size_t)(ov - 3);
            fs->rep[2] = fs->rep[1];
            fs->rep[1] = fs->rep[0];
            fs->rep[0] = (uint32_t)offset;
          } else {
            uint32_t idx = (uint32_t)ov - 1 + (ll == 0 ? 1u : 0u); /* 0..3 */
            if (idx == 0) {
              offset = fs->rep[0];
            } else {
              offset = idx == 3 ? (size_t)fs->rep[0] - 1 : fs->rep[idx];
              if (idx > 1) fs->rep[2] = fs->rep[1];
              fs->rep[1] = fs->rep[0];
              fs->rep[0] = (uint32_t)offset;
            }
          }
          if (ll > regen - lit_at || ll > cap - *op) {
            st = ZERR;
            break;
          }
          memcpy(out + *op, lit + lit_at, ll);
          *op += ll;
          lit_at += ll;
          st = copy_match(out, op, cap, offset, ml);
        }
        if (!st && b.bit != 0) st = ZERR;
      }
    }
  } else if (!st && pos != n) {
    st = ZERR; /* no sequences: the section is that one byte */
  }
  if (!st) {
    const size_t rest = regen - lit_at;
    if (rest > cap - *op) st = ZERR;
    else {
      memcpy(out + *op, lit + lit_at, rest);
      *op += rest;
    }
  }
  free(lit);
  return st;
}

/* ---- XXH64 (content checksum) ---- */
#define P1 0x9E3779B185EBCA87ULL
#define P2 0xC2B2AE3D27D4EB4FULL
#define P3 0x165667B19E3779F9ULL
#define P4 0x85EBCA77C2B2AE63ULL
#define P5 0x27D4EB2F165667C5ULL
static uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t rd64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}
static uint64_t xround(uint64_t acc, uint64_t in) { return rotl(acc + in * P2, 31) * P1; }
static uint64_t xmerge(uint64_t acc, uint64_t v) { return (acc ^ xround(0, v)) * P1 + P4; }
uint64_t orc_xxh64(const uint8_t* p, size_t n) {
  const uint8_t* end = p + n;
  uint64_t h;
  if (n >= 32) {
    uint64_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0 - P1;
    do {
      v1 = xround(v1, rd64(p));
      v2 = xround(v2, rd64(p + 8));
      v3 = xround(v3, rd64(p + 16));
      v4 = xround(v4, rd64(p + 24));
      p += 32;
    } while (p + 32 <= end);
    h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    h = xmerge(h, v1);
    h = xmerge(h, v2);
    h = xmerge(h, v3);
    h = xmerge(h, v4);
  } else {
    h = P5;
  }
  h += (uint64_t)n;
  while (p + 8 <= end) {
    h ^= xround(0, rd64(p));
    h = rotl(h, 27) * P1 + P4;
    p += 8;
  }
  if (p + 4 <= end) {
    uint32_t v;
    memcpy(&v, p, 4);
    h ^= (uint64_t)v * P1;
    h = rotl(h, 23) * P2 + P3;
    p += 4;
  }
  while (p < end) {
    h ^= (uint64_t)(*p++) * P5;
    h = rotl(h, 11) * P1;
  }
  h ^= h >> 33;
  h *= P2;
  h ^= h >> 29;
  h *= P3;
  h ^= h >> 32;
  return h;
}

int orc_zstd_decompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t cap, size_t* produced, size_t* consumed) {
  size_t ip = 0, op = 0;
  int st = ORC_OK;
  *produced = *consumed = 0;
  FrameState* fs = (FrameState*)calloc(1, sizeof(FrameState));
  if (!fs) return ZERR;
  while (ip < in_len && !st) { /* frames one behind the other */
    if (in_len - ip < 4) {
      st = ZERR;
      break;
    }
    uint32_t magic;
    memcpy(&magic, in + ip, 4);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) { /* skippable frame */
      if (in_len - ip < 8) {
        st = ZERR;
        break;
      }
      uint32_t sz;
      memcpy(&sz, in + ip + 4, 4);
      if (in_len - ip - 8 < sz) {
        st = ZERR;
        break;
      }
      ip += 8 + (size_t)sz;
      continue;
    }
    if (magic != 0xFD2FB528u || in_len - ip < 5) {
      st = ZERR;
      break;
    }
    const uint8_t fhd = in[ip + 4];
    size_t p = ip + 5;
    const int single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, did = fhd & 3, fcs = fhd >> 6;
    if (fhd & 0x08) {
      st = ZERR;
      break;
    }
    const size_t need = (single ? 0 : 1) + (did == 3 ? 4 : (size_t)did) + (fcs == 0 ? (size_t)single : (size_t)1 << fcs);
    if (in_len - p < need) {
      st = ZERR;
      break;
    }
    if (!single) ++p; /* window size: no limit is enforced here */
    uint32_t dict_id = 0;
    for (size_t i = 0; i < (did == 3 ? 4u : (size_t)did); ++i) dict_id |= (uint32_t)in[p++] << (8 * i);
    if (dict_id != 0) {
      st = ZERR; /* a dictionary is needed */
      break;
    }
    uint64_t content = 0;
    int have_content = 0;
    {
      const size_t nb = fcs == 0 ? (size_t)single : (size_t)1 << fcs;
      for (size_t i = 0; i < nb; ++i) content |= (uint64_t)in[p++] << (8 * i);
      if (nb == 2) content += 256;
      have_content = nb != 0;
    }
    memset(fs, 0, sizeof *fs);
    fs->rep[0] = 1;
    fs->rep[1] = 4;
    fs->rep[2] = 8;
    const size_t frame_start = op;
    for (;;) {
      if (in_len - p < 3) {
        st = ZERR;
        break;
      }
      const uint32_t bh = in[p] | ((uint32_t)in[p + 1] << 8) | ((uint32_t)in[p + 2] << 16);
      p += 3;
      const int last = bh & 1, type = (bh >> 1) & 3;
      const size_t size = bh >> 3;
      if (type == 0) {
        if (in_len - p < size || size > cap - op) {
          st = ZERR;
          break;
        }
        memcpy(out + op, in + p, size);
        op += size;
        p += size;
      } else if (type == 1) {
        if (in_len - p < 1 || size > cap - op) {
          st = ZERR;
          break;
        }
        memset(out + op, in[p], size);
        op += size;
        p += 1;
      } else if (type == 2) {
        if (in_len - p < size || size > 128 * 1024) {
          st = ZERR;
          break;
        }
        /* matches reach back into the frame only: the block decoder sees the frame's output so far */
        size_t fop = op - frame_start;
        st = compressed_block(fs, in + p, size, out + frame_start, &fop, cap - frame_start);
        op = frame_start + fop;
        if (st) break;
        p += size;
      } else {
        st = ZERR;
        break;
      }
      if (last) break;
    }
    if (st) break;
    if (have_content && content != op - frame_start) {
      st = ZERR;
      break;
    }
    if (checksum) {
      if (in_len - p < 4) {
        st = ZERR;
        break;
      }
      uint32_t want;
      memcpy(&want, in + p, 4);
      if ((uint32_t)orc_xxh64(out + frame_start, op - frame_start) != want) {
        st = ZERR;
        break;
      }
      p += 4;
    }
    ip = p;
  }
  if (!st && op != cap) st = ZERR; /* the page header announced cap bytes */
  free(fs);
  *produced = op;
  *consumed = ip;
  return st;
}
