/*
 * TEST INFRASTRUCTURE ONLY -- scalar CPU restatement of the 128-integer block
 * bit-packing codec of lemire/simdcomp.  It is the checker the HIP kernels are
 * compared against; it is never linked into, loaded by or called from the
 * shipped library (libsimdcomp_hip.so).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it.
 *
 * Parity status: PINNED.  The reference holds no golden vectors of its own
 * (its tests are round trips), so this restatement is pinned against outputs of
 * the reference itself, compiled from /root/reference by oracle/Makefile into
 * oracle/_ref/, and against tests/golden/ fixtures generated from that build
 * (tools/gen_golden.py).
 *
 * Everything is written as plain per-value arithmetic on the wire format
 * rather than as unrolled vector code:
 *
 *   value i of a block -> lane = i & 3, row k = i >> 2
 *   lane bit position p = k*b, word w = p >> 5, shift s = p & 31
 *   compressed word index 4*w + lane (uint32, little endian)
 *
 * Each function cites the reference routine (file:line under /root/reference)
 * whose observable behaviour it restates.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define ORC_BLOCK 128u

static uint32_t field_mask(uint32_t b) {
  return b >= 32 ? 0xFFFFFFFFu : ((1u << b) - 1u);
}

/* src/simdcomputil.c:16 bits() */
uint32_t orc_bits(uint32_t v) {
  uint32_t n = 0;
  while (v) {
    ++n;
    v >>= 1;
  }
  return n;
}

/* src/simdcomputil.c:160 maxbits_length(); :42 maxbits() is length == 128 */
uint32_t orc_maxbits_length(const uint32_t *in, uint32_t length) {
  uint32_t acc = 0, i;
  for (i = 0; i < length; ++i) acc |= in[i];
  return orc_bits(acc);
}

uint32_t orc_maxbits(const uint32_t *in) { return orc_maxbits_length(in, ORC_BLOCK); }

/* src/simdcomputil.c:198 simdmaxbitsd1_length(): OR of successive differences
 * seeded with initvalue.  The reference pads lengths 1..3 by repeating the last
 * value (:213-226), which only adds zero differences. length == 0 trips an
 * assert there (:209); here it returns 0. */
uint32_t orc_simdmaxbitsd1_length(uint32_t initvalue, const uint32_t *in, uint32_t length) {
  uint32_t acc = 0, prev = initvalue, i;
  for (i = 0; i < length; ++i) {
    acc |= in[i] - prev;
    prev = in[i];
  }
  return orc_bits(acc);
}

/* src/simdcomputil.c:181 simdmaxbitsd1() */
uint32_t orc_simdmaxbitsd1(uint32_t initvalue, const uint32_t *in) {
  return orc_simdmaxbitsd1_length(initvalue, in, ORC_BLOCK);
}

/* src/simdcomputil.c:130 simdmaxmin_length(): unsigned min and max; an empty
 * range yields min = 0xFFFFFFFF, max = 0 (:135-136). */
void orc_simdmaxmin_length(const uint32_t *in, uint32_t length, uint32_t *getmin, uint32_t *getmax) {
  uint32_t lo = 0xFFFFFFFFu, hi = 0, i;
  for (i = 0; i < length; ++i) {
    if (in[i] < lo) lo = in[i];
    if (in[i] > hi) hi = in[i];
  }
  *getmin = lo;
  *getmax = hi;
}

/* src/simdcomputil.c:95 simdmaxmin() */
void orc_simdmaxmin(const uint32_t *in, uint32_t *getmin, uint32_t *getmax) {
  orc_simdmaxmin_length(in, ORC_BLOCK, getmin, getmax);
}

/* src/simdcomputil.c:109 simdmin_length() and :84 simdmin() */
uint32_t orc_simdmin_length(const uint32_t *in, uint32_t length) {
  uint32_t lo, hi;
  orc_simdmaxmin_length(in, length, &lo, &hi);
  return lo;
}

uint32_t orc_simdmin(const uint32_t *in) { return orc_simdmin_length(in, ORC_BLOCK); }

/* ------------------------------------------------------------------------ */
/* One block.  `x` are the 128 values actually fed to the packer (after any   */
/* delta / frame subtraction); out receives exactly b vectors of 16 bytes.    */
/* ------------------------------------------------------------------------ */

/* Block packing rule shared by src/simdbitpacking.c:13913 simdpackwithoutmask,
 * :14052 simdpack, src/simdintegratedbitpacking.c:25058/25198 and
 * src/simdfor.c:14705: value << s ORed into word w, and when the field crosses
 * the word (s + b > 32) the whole of value >> (32 - s) ORed into word w + 1.
 * Nothing is masked here, so out-of-range inputs smear exactly as they do in
 * the reference's unmasked kernels. */
static void block_pack_raw(const uint32_t *x, uint32_t *out, uint32_t b) {
  uint32_t i;
  if (b == 0) return;
  if (b >= 32) {
    memcpy(out, x, ORC_BLOCK * sizeof(uint32_t));
    return;
  }
  memset(out, 0, 16u * b);
  for (i = 0; i < ORC_BLOCK; ++i) {
    uint32_t lane = i & 3u, p = (i >> 2) * b, w = p >> 5, s = p & 31u;
    out[4 * w + lane] |= x[i] << s;
    if (s + b > 32) out[4 * (w + 1) + lane] |= x[i] >> (32 - s);
  }
}

/* Block extraction rule of src/simdbitpacking.c:13773 simdunpack: b == 0 gives
 * zeros (:6-9), b == 32 a copy (:13764-13771). */
static void block_unpack_raw(const uint32_t *in, uint32_t *v, uint32_t b) {
  uint32_t i, m = field_mask(b);
  if (b == 0) {
    memset(v, 0, ORC_BLOCK * sizeof(uint32_t));
    return;
  }
  if (b >= 32) {
    memcpy(v, in, ORC_BLOCK * sizeof(uint32_t));
    return;
  }
  for (i = 0; i < ORC_BLOCK; ++i) {
    uint32_t lane = i & 3u, p = (i >> 2) * b, w = p >> 5, s = p & 31u;
    uint32_t f = in[4 * w + lane] >> s;
    if (s + b > 32) f |= in[4 * (w + 1) + lane] << (32 - s);
    v[i] = f & m;
  }
}

/* src/simdbitpacking.c:13913 simdpackwithoutmask(); bit > 32 does nothing (:14046) */
void orc_simdpackwithoutmask(const uint32_t *in, void *out, uint32_t bit) {
  if (bit > 32) return;
  block_pack_raw(in, (uint32_t *)out, bit);
}

/* src/simdbitpacking.c:14052 simdpack(): inputs masked to b bits first (:4412-4414) */
void orc_simdpack(const uint32_t *in, void *out, uint32_t bit) {
  uint32_t x[ORC_BLOCK], i, m = field_mask(bit);
  if (bit > 32) return;
  for (i = 0; i < ORC_BLOCK; ++i) x[i] = in[i] & m;
  block_pack_raw(x, (uint32_t *)out, bit);
}

/* src/simdbitpacking.c:13773 simdunpack() */
void orc_simdunpack(const void *in, uint32_t *out, uint32_t bit) {
  if (bit > 32) return;
  block_unpack_raw((const uint32_t *)in, out, bit);
}

/* The 8-lane and 16-lane wire formats of src/avxbitpacking.c:9906-9918
 * (avxpack / avxpackwithoutmask / avxunpack, 256 integers, AVX2) and
 * src/avx512bitpacking.c (avx512pack / ..., 512 integers): the same rule with
 * `lanes` interleaved lanes -- value i in lane i % lanes, row i / lanes, word w
 * of lane l at uint32 index lanes*w + l; a block is 32*lanes integers and
 * `bit` vectors of 4*lanes bytes.  lanes = 4 is the SSE format above. */
void orc_lanes_pack(uint32_t lanes, int masked, const uint32_t *in, void *outv, uint32_t bit) {
  uint32_t *out = (uint32_t *)outv;
  uint32_t i, n = 32 * lanes, m = masked ? field_mask(bit) : 0xFFFFFFFFu;
  if (bit == 0 || bit > 32) return;
  if (bit == 32) {
    memcpy(out, in, n * sizeof(uint32_t));
    return;
  }
  memset(out, 0, 4u * lanes * bit);
  for (i = 0; i < n; ++i) {
    uint32_t lane = i % lanes, p = (i / lanes) * bit, w = p >> 5, s = p & 31u, v = in[i] & m;
    out[lanes * w + lane] |= v << s;
    if (s + bit > 32) out[lanes * (w + 1) + lane] |= v >> (32 - s);
  }
}

void orc_lanes_unpack(uint32_t lanes, const void *inv, uint32_t *v, uint32_t bit) {
  const uint32_t *in = (const uint32_t *)inv;
  uint32_t i, n = 32 * lanes, m = field_mask(bit);
  if (bit > 32) return;
  if (bit == 0) {
    memset(v, 0, n * sizeof(uint32_t));
    return;
  }
  if (bit == 32) {
    memcpy(v, in, n * sizeof(uint32_t));
    return;
  }
  for (i = 0; i < n; ++i) {
    uint32_t lane = i % lanes, p = (i / lanes) * bit, w = p >> 5, s = p & 31u;
    uint32_t f = in[lanes * w + lane] >> s;
    if (s + bit > 32) f |= in[lanes * (w + 1) + lane] << (32 - s);
    v[i] = f & m;
  }
}

/* avxmaxbits (src/avxbitpacking.c:17) / avx512maxbits */
uint32_t orc_lanes_maxbits(uint32_t lanes, const uint32_t *in) { return orc_maxbits_length(in, 32 * lanes); }

/* src/simdintegratedbitpacking.c:25058 simdpackwithoutmaskd1(): successive
 * differences seeded by initvalue (Delta, :6-12), packed unmasked.  b == 32
 * stores the raw values and ignores initvalue (:15978-15985). */
void orc_simdpackwithoutmaskd1(uint32_t initvalue, const uint32_t *in, void *out, uint32_t bit) {
  uint32_t d[ORC_BLOCK], i, prev = initvalue;
  if (bit > 32) return;
  if (bit == 32) {
    memcpy(out, in, ORC_BLOCK * sizeof(uint32_t));
    return;
  }
  for (i = 0; i < ORC_BLOCK; ++i) {
    d[i] = in[i] - prev;
    prev = in[i];
  }
  block_pack_raw(d, (uint32_t *)out, bit);
}

/* src/simdintegratedbitpacking.c:25198 simdpackd1(): as above with each
 * difference masked to b bits (:1062); b == 32 raw copy (:16205-16212). */
void orc_simdpackd1(uint32_t initvalue, const uint32_t *in, void *out, uint32_t bit) {
  uint32_t d[ORC_BLOCK], i, prev = initvalue, m = field_mask(bit);
  if (bit > 32) return;
  if (bit == 32) {
    memcpy(out, in, ORC_BLOCK * sizeof(uint32_t));
    return;
  }
  for (i = 0; i < ORC_BLOCK; ++i) {
    d[i] = (in[i] - prev) & m;
    prev = in[i];
  }
  block_pack_raw(d, (uint32_t *)out, bit);
}

/* src/simdintegratedbitpacking.c:24915 simdunpackd1(): extraction followed by a
 * running sum seeded with initvalue (PrefixSum, :14-19).  b == 0 yields
 * initvalue 128 times (:21-35); b == 32 is a raw copy with no sum
 * (:24903-24913). */
void orc_simdunpackd1(uint32_t initvalue, const void *in, uint32_t *out, uint32_t bit) {
  uint32_t i, run = initvalue;
  if (bit > 32) return;
  block_unpack_raw((const uint32_t *)in, out, bit);
  if (bit == 32) return;
  for (i = 0; i < ORC_BLOCK; ++i) {
    run += out[i];
    out[i] = run;
  }
}

/* src/simdfor.c:14705 simdpackFOR(): value - base (wrapping, :33) packed
 * unmasked; b == 32 raw copy (:6954-6962). */
void orc_simdpackFOR(uint32_t initvalue, const uint32_t *in, void *out, uint32_t bit) {
  uint32_t x[ORC_BLOCK], i;
  if (bit > 32) return;
  if (bit == 32) {
    memcpy(out, in, ORC_BLOCK * sizeof(uint32_t));
    return;
  }
  for (i = 0; i < ORC_BLOCK; ++i) x[i] = in[i] - initvalue;
  block_pack_raw(x, (uint32_t *)out, bit);
}

/* src/simdfor.c:14846 simdunpackFOR(): extraction + base (:7539); b == 0 gives
 * base 128 times (:7-20); b == 32 raw copy (:14692-14703). */
void orc_simdunpackFOR(uint32_t initvalue, const void *in, uint32_t *out, uint32_t bit) {
  uint32_t i;
  if (bit > 32) return;
  block_unpack_raw((const uint32_t *)in, out, bit);
  if (bit == 32) return;
  for (i = 0; i < ORC_BLOCK; ++i) out[i] += initvalue;
}

/* ------------------------------------------------------------------------ */
/* Arbitrary lengths                                                          */
/* ------------------------------------------------------------------------ */

/* src/simdbitpacking.c:14370 simdpack_compressedbytes() and
 * src/simdfor.c:15034 simdpackFOR_compressedbytes(): int arithmetic there;
 * 64-bit here so the large-array callers can use it too. */
int64_t orc_compressedbytes(int64_t length, uint32_t bit) {
  if (bit == 0) return 0;
  if (bit == 32) return length * 4;
  return (((length + 3) / 4) * (int64_t)bit + 31) / 32 * 16;
}

int orc_simdpack_compressedbytes(int length, uint32_t bit) { return (int)orc_compressedbytes(length, bit); }
int orc_simdpackFOR_compressedbytes(int length, uint32_t bit) { return (int)orc_compressedbytes(length, bit); }

/* The "generic loop" packer of src/simdbitpacking.c:14190 simdpack_shortlength()
 * and src/simdfor.c:15043 simdpackFOR_length(), restated per lane: each of the
 * four lanes is an independent bit stream; `acc` is the word under
 * construction and `fill` how many of its bits are taken.  A value is ORed in
 * at `fill`; if it does not leave room for another bit (b >= 32 - fill) the
 * word is emitted and the next word starts with value >> (32 - fill).  Note
 * the >= : when a field ends exactly on the word boundary the next word still
 * starts with value >> b, which is zero only for in-range values
 * (simdbitpacking.c:14208-14214).  The final partial word is emitted only if it
 * holds bits (:14236).  Values beyond `length` in the last row are `pad`
 * (0 at simdbitpacking.c:14222, the frame base at simdfor.c:15078-15080) and
 * `sub` is subtracted from every value (0 for the plain packer).
 * Returns the number of 16-byte vectors written. */
static size_t generic_pack(const uint32_t *in, size_t length, uint32_t *out, uint32_t b, uint32_t sub,
                           uint32_t pad) {
  size_t rows = (length + 3) / 4, k, nvec = 0;
  uint32_t lane;
  for (lane = 0; lane < 4; ++lane) {
    uint32_t acc = 0, fill = 0;
    size_t w = 0;
    for (k = 0; k < rows; ++k) {
      size_t idx = 4 * k + lane;
      uint32_t v = (idx < length ? in[idx] : pad) - sub;
      acc |= v << fill;
      if (b < 32 - fill) {
        fill += b;
      } else {
        uint32_t room = 32 - fill;
        out[4 * w + lane] = acc;
        ++w;
        acc = v >> room;
        fill = b - room;
      }
    }
    if (fill != 0) {
      out[4 * w + lane] = acc;
      ++w;
    }
    nvec = w;
  }
  return nvec;
}

/* src/simdbitpacking.c:14190 simdpack_shortlength(): returns the advanced
 * output pointer; b == 32 copies length*4 bytes and advances by length uint32
 * (:14198-14201). */
void *orc_simdpack_shortlength(const uint32_t *in, int length, void *out, uint32_t bit) {
  if (bit == 0) return out;
  if (bit == 32) {
    memcpy(out, in, (size_t)length * 4);
    return (uint32_t *)out + length;
  }
  return (uint8_t *)out + 16 * generic_pack(in, (size_t)length, (uint32_t *)out, bit, 0, 0);
}

/* Extraction for arbitrary length used by src/simdbitpacking.c:14242
 * simdunpack_shortlength() and src/simdfor.c:15098 simdunpackFOR_length():
 * only the vectors that hold bits of the first `length` values are read. */
static size_t generic_unpack(const uint32_t *in, size_t length, uint32_t *out, uint32_t b, uint32_t add) {
  size_t i;
  uint32_t m = field_mask(b);
  for (i = 0; i < length; ++i) {
    uint32_t lane = (uint32_t)(i & 3u);
    uint64_t p = (uint64_t)(i >> 2) * b;
    size_t w = (size_t)(p >> 5);
    uint32_t s = (uint32_t)(p & 31u);
    uint32_t f = in[4 * w + lane] >> s;
    if (s + b > 32) f |= in[4 * (w + 1) + lane] << (32 - s);
    out[i] = (f & m) + add;
  }
  return (size_t)((((uint64_t)((length + 3) / 4)) * b + 31) / 32);
}

/* src/simdbitpacking.c:14242 simdunpack_shortlength() */
const void *orc_simdunpack_shortlength(const void *in, int length, uint32_t *out, uint32_t bit) {
  int k;
  if (length == 0) return in;
  if (bit == 0) {
    for (k = 0; k < length; ++k) out[k] = 0;
    return in;
  }
  if (bit == 32) {
    memcpy(out, in, (size_t)length * 4);
    return (const uint32_t *)in + length;
  }
  return (const uint8_t *)in + 16 * generic_unpack((const uint32_t *)in, (size_t)length, out, bit, 0);
}

/* src/simdbitpacking.c:14379 simdpack_length(): whole blocks through the MASKED
 * block packer, the remainder through the unmasked short packer. */
void *orc_simdpack_length(const uint32_t *in, size_t length, void *out, uint32_t bit) {
  size_t k, nblocks = length / ORC_BLOCK;
  uint8_t *o = (uint8_t *)out;
  for (k = 0; k < nblocks; ++k) {
    orc_simdpack(in, o, bit);
    in += ORC_BLOCK;
    o += 16u * bit;
  }
  return orc_simdpack_shortlength(in, (int)(length % ORC_BLOCK), o, bit);
}

/* src/simdbitpacking.c:14390 simdunpack_length() */
const void *orc_simdunpack_length(const void *in, size_t length, uint32_t *out, uint32_t bit) {
  size_t k, nblocks = length / ORC_BLOCK;
  const uint8_t *p = (const uint8_t *)in;
  for (k = 0; k < nblocks; ++k) {
    orc_simdunpack(p, out, bit);
    out += ORC_BLOCK;
    p += 16u * bit;
  }
  return orc_simdunpack_shortlength(p, (int)(length % ORC_BLOCK), out, bit);
}

/* src/simdfor.c:15043 simdpackFOR_length(): the whole array goes through the
 * generic loop (not the block kernels), last row padded with the base. */
void *orc_simdpackFOR_length(uint32_t initvalue, const uint32_t *in, int length, void *out, uint32_t bit) {
  if (bit == 0) return out;
  if (bit == 32) {
    memcpy(out, in, (size_t)length * 4);
    return (uint32_t *)out + length;
  }
  return (uint8_t *)out + 16 * generic_pack(in, (size_t)length, (uint32_t *)out, bit, initvalue, initvalue);
}

/* src/simdfor.c:15098 simdunpackFOR_length() */
const void *orc_simdunpackFOR_length(uint32_t initvalue, const void *in, int length, uint32_t *out,
                                     uint32_t bit) {
  int k;
  if (length == 0) return in;
  if (bit == 0) {
    for (k = 0; k < length; ++k) out[k] = initvalue;
    return in;
  }
  if (bit == 32) {
    memcpy(out, in, (size_t)length * 4);
    return (const uint32_t *)in + length;
  }
  return (const uint8_t *)in + 16 * generic_unpack((const uint32_t *)in, (size_t)length, out, bit, initvalue);
}

/* ------------------------------------------------------------------------ */
/* Single-value access into one packed block                                  */
/* ------------------------------------------------------------------------ */

/* src/simdfor.c:14987 simdselectFOR() */
uint32_t orc_simdselectFOR(uint32_t initvalue, const void *in, uint32_t bit, int slot) {
  const uint32_t *w = (const uint32_t *)in;
  uint32_t lane, p, s, f;
  if (bit == 0) return initvalue;
  if (bit == 32) return w[slot];
  lane = (uint32_t)slot & 3u;
  p = ((uint32_t)slot >> 2) * bit;
  s = p & 31u;
  f = w[4 * (p >> 5) + lane] >> s;
  if (s + bit > 32) f |= w[4 * ((p >> 5) + 1) + lane] << (32 - s);
  return initvalue + (f & field_mask(bit));
}

/* src/simdfor.c:15016 simdsearchwithlengthFOR(): lower bound by bisection */
int orc_simdsearchwithlengthFOR(uint32_t initvalue, const void *in, uint32_t bit, int length, uint32_t key,
                                uint32_t *presult) {
  int lo = 0, n = length;
  while (n > 0) {
    int half = n / 2;
    if (orc_simdselectFOR(initvalue, in, bit, lo + half) < key) {
      lo += half + 1;
      n -= half + 1;
    } else {
      n = half;
    }
  }
  *presult = orc_simdselectFOR(initvalue, in, bit, lo);
  return lo;
}

/* src/simdpackedselect.c:8096 simdselectd1(): the slot-th value of the
 * differentially coded block; the slot is reduced mod 128 (":8100 slot &= 127"),
 * a 0-bit block yields the offset, widths above 32 yield (uint32_t)-1 (:8239). */
uint32_t orc_simdselectd1(uint32_t initvalue, const void *in, uint32_t bit, int slot) {
  uint32_t tmp[ORC_BLOCK];
  if (bit > 32) return (uint32_t)-1;
  orc_simdunpackd1(initvalue, in, tmp, bit);
  return tmp[slot & 127];
}

/* The bisection the reference uses on raw 32-bit blocks
 * (src/simdpackedsearch.c:29 lower_bound): not std::lower_bound on unsorted
 * input, so it is restated step for step. */
static int ref_lower_bound(const uint32_t *a, uint32_t key, int imin, int imax) {
  imax--;
  while (imin + 1 < imax) {
    int imid = imin + (imax - imin) / 2;
    if (a[imid] >= key)
      imax = imid;
    else
      imin = imid;
  }
  return a[imin] >= key ? imin : imax;
}

/* src/simdpackedsearch.c:8284 simdsearchwithlengthd1(): position of the first
 * of the first `length` decoded values that is >= key (unsigned), its value in
 * *presult; `length` and key + 1 when there is none.  Widths 1..31 scan four
 * values at a time (:58-73), which amounts to a linear scan; width 0 tests the
 * offset once (:76-89); width 32 bisects the raw values (:8270-8282).  The
 * reference reads in32[-1] for width 32 with length 0; here that case is "not found". */
int orc_simdsearchwithlengthd1(uint32_t initvalue, const void *in, uint32_t bit, int length, uint32_t key,
                               uint32_t *presult) {
  uint32_t tmp[ORC_BLOCK];
  int i;
  if (bit > 32) return -1;
  if (bit == 32) {
    const uint32_t *raw = (const uint32_t *)in;
    int answer;
    if (length <= 0) {
      *presult = key + 1;
      return length;
    }
    answer = ref_lower_bound(raw, key, 0, length);
    if (raw[answer] < key) {
      *presult = key + 1;
      return length;
    }
    *presult = raw[answer];
    return answer;
  }
  if (bit == 0) {
    if (length > 0 && initvalue >= key) {
      *presult = initvalue;
      return 0;
    }
    *presult = key + 1;
    return length;
  }
  orc_simdunpackd1(initvalue, in, tmp, bit);
  for (i = 0; i < length && i < (int)ORC_BLOCK; ++i) {
    if (tmp[i] >= key) {
      *presult = tmp[i];
      return i;
    }
  }
  *presult = key + 1;
  return length;
}

/* src/simdpackedsearch.c:16583 simdsearchd1(): the same over all 128 values,
 * with the running offset kept as a vector of the last four decoded values
 * that the caller carries from block to block: on return it holds the group of
 * four in which the search stopped (the last group when nothing was found);
 * width 0 leaves it alone (:8410-8420), width 32 loads the block's last vector
 * only when the key was found (:16570-16581).  Only element 3 seeds the block. */
int orc_simdsearchd1(uint32_t offset4[4], const void *in, uint32_t bit, uint32_t key, uint32_t *presult) {
  uint32_t tmp[ORC_BLOCK];
  int pos, g;
  if (bit > 32) return -1;
  pos = orc_simdsearchwithlengthd1(offset4[3], in, bit, (int)ORC_BLOCK, key, presult);
  if (bit == 0) return pos;
  if (bit == 32) {
    if (pos < (int)ORC_BLOCK) memcpy(offset4, (const uint32_t *)in + 124, 16);
    return pos;
  }
  orc_simdunpackd1(offset4[3], in, tmp, bit);
  g = pos < (int)ORC_BLOCK ? pos / 4 : 31;
  memcpy(offset4, tmp + 4 * g, 16);
  return pos;
}

/* src/simdpackedselect.c:15216 simdscand1(): walks the block and leaves its last
 * four values in the offset vector (width 0: untouched). */
void orc_simdscand1(uint32_t offset4[4], const void *in, uint32_t bit) {
  uint32_t tmp[ORC_BLOCK];
  if (bit == 0 || bit > 32) return;
  orc_simdunpackd1(offset4[3], in, tmp, bit);
  memcpy(offset4, tmp + 124, 16);
}

/* src/simdbitpacking.c:14338 simdfastset(): overwrite one field in place.  The
 * low part is cleared with the b-bit mask and the raw value ORed in; the
 * spill word has its low (s + b - 32) bits replaced by value >> (32 - s). */
void orc_simdfastset(void *in128, uint32_t b, uint32_t value, size_t index) {
  uint32_t *w = (uint32_t *)in128;
  uint32_t lane = (uint32_t)(index & 3u), p = (uint32_t)(index >> 2) * b, s = p & 31u;
  uint32_t *first = w + 4 * (p >> 5) + lane;
  if (b == 0) return;
  if (b == 32)
    *first = 0;
  else
    *first &= ~(field_mask(b) << s);
  *first |= value << s;
  if (s + b > 32) {
    uint32_t keep = s + b - 32;
    first[4] &= ~field_mask(keep);
    first[4] |= value >> (32 - s);
  }
}

/* src/simdfor.c:15197 simdfastsetFOR() */
void orc_simdfastsetFOR(uint32_t initvalue, void *in, uint32_t bit, uint32_t value, size_t index) {
  orc_simdfastset(in, bit, value - initvalue, index);
}

/* ------------------------------------------------------------------------ */
/* Batched drivers used by the tests and by bench.py's CPU leg: one call per  */
/* array instead of one ctypes call per block.  mode: 0 plain, 1 d1, 2 FOR.   */
/* Widths: bits[k] if bits != NULL else `bit`; payload of block k starts at   */
/* byte off[k] if off != NULL else k*16*bit; seeds[k] is the d1 offset / FOR  */
/* base of block k.                                                           */
/* ------------------------------------------------------------------------ */

void orc_batch_pack(int mode, int masked, const uint32_t *seeds, const uint32_t *in, uint8_t *out,
                    const uint64_t *off, const uint8_t *bits, uint32_t bit, size_t nblocks) {
  size_t k;
  for (k = 0; k < nblocks; ++k) {
    uint32_t b = bits ? bits[k] : bit;
    uint8_t *o = out + (off ? off[k] : (uint64_t)k * 16u * bit);
    const uint32_t *x = in + k * ORC_BLOCK;
    if (mode == 0) {
      if (masked) orc_simdpack(x, o, b); else orc_simdpackwithoutmask(x, o, b);
    } else if (mode == 1) {
      if (masked) orc_simdpackd1(seeds[k], x, o, b); else orc_simdpackwithoutmaskd1(seeds[k], x, o, b);
    } else {
      orc_simdpackFOR(seeds[k], x, o, b);
    }
  }
}

void orc_batch_unpack(int mode, const uint32_t *seeds, const uint8_t *in, const uint64_t *off,
                      const uint8_t *bits, uint32_t bit, uint32_t *out, size_t nblocks) {
  size_t k;
  for (k = 0; k < nblocks; ++k) {
    uint32_t b = bits ? bits[k] : bit;
    const uint8_t *p = in + (off ? off[k] : (uint64_t)k * 16u * bit);
    uint32_t *o = out + k * ORC_BLOCK;
    if (mode == 0) orc_simdunpack(p, o, b);
    else if (mode == 1) orc_simdunpackd1(seeds[k], p, o, b);
    else orc_simdunpackFOR(seeds[k], p, o, b);
  }
}

/* per-block reductions: what 0 maxbits, 1 maxbitsd1 (seeds[k] = offset), 2 min/max */
void orc_batch_reduce(int what, const uint32_t *seeds, const uint32_t *in, size_t nblocks, uint8_t *bits_out,
                      uint32_t *min_out, uint32_t *max_out) {
  size_t k;
  for (k = 0; k < nblocks; ++k) {
    const uint32_t *x = in + k * ORC_BLOCK;
    if (what == 0) bits_out[k] = (uint8_t)orc_maxbits(x);
    else if (what == 1) bits_out[k] = (uint8_t)orc_simdmaxbitsd1(seeds[k], x);
    else orc_simdmaxmin(x, &min_out[k], &max_out[k]);
  }
}

/* 64-bit FNV-1a over a byte range: the digest the golden fixtures store for
 * corpora too large to commit. */
uint64_t orc_fnv1a64(const void *data, size_t nbytes) {
  const uint8_t *p = (const uint8_t *)data;
  uint64_t h = 0xcbf29ce484222325ull;
  size_t i;
  for (i = 0; i < nbytes; ++i) {
    h ^= p[i];
    h *= 0x100000001b3ull;
  }
  return h;
}
