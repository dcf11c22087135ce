// Host-side routines of the reference API that touch at most one packed value
// or none at all: size arithmetic, bit width of a scalar, and the random-access
// helpers that read or overwrite a single field of one packed block in host
// memory.  They have no batch dimension to put on a GPU and are plain C++.
// (The data-parallel routines are in host_api.hip / device_api.hip.)
#include <stddef.h>
#include <stdint.h>

#include "../../include/simdcomp.h"

namespace {

struct Field {
  size_t word;     // uint32 index of the word holding the field's low bits
  uint32_t shift;  // bit position inside that word
  bool split;      // the field continues in the same lane's next word (4 uint32 further)
};

// Lane = slot mod 4, row = slot / 4; a lane's rows are laid end to end, b bits each
// (src/simdfor.c:14996-14999, src/simdbitpacking.c:14340-14343).
inline Field locate(size_t slot, uint32_t b) {
  const uint32_t pos = (uint32_t)(slot >> 2) * b;
  Field f;
  f.word = 4 * (size_t)(pos >> 5) + (slot & 3);
  f.shift = pos & 31;
  f.split = f.shift + b > 32;
  return f;
}

inline uint32_t low_ones(uint32_t n) { return n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u); }

}  // namespace

extern "C" {

// src/simdcomputil.c:16
uint32_t bits(const uint32_t v) { return v ? 32u - (uint32_t)__builtin_clz(v) : 0u; }

// src/simdbitpacking.c:14370 (int arithmetic, as in the reference)
int simdpack_compressedbytes(int length, const uint32_t bit) {
  if (bit == 0) return 0;
  if (bit == 32) return length * (int)sizeof(uint32_t);
  return (((length + 3) / 4) * (int)bit + 31) / 32 * 16;
}

// src/simdfor.c:15034
int simdpackFOR_compressedbytes(int length, const uint32_t bit) { return simdpack_compressedbytes(length, bit); }

// src/simdfor.c:14987
uint32_t simdselectFOR(uint32_t initvalue, const __m128i *in, uint32_t bit, int slot) {
  const uint32_t *w = reinterpret_cast<const uint32_t *>(in);
  if (bit == 0) return initvalue;
  if (bit == 32) return w[slot];
  const Field f = locate((size_t)slot, bit);
  uint32_t v = w[f.word] >> f.shift;
  if (f.split) v |= w[f.word + 4] << (32 - f.shift);
  return initvalue + (v & low_ones(bit));
}

// src/simdfor.c:15016: first slot whose value is >= key among the first `length`
int simdsearchwithlengthFOR(uint32_t initvalue, const __m128i *in, uint32_t bit, int length, uint32_t key,
                            uint32_t *presult) {
  int lo = 0, span = length;
  while (span > 0) {
    const int half = span >> 1;
    if (simdselectFOR(initvalue, in, bit, lo + half) < key) {
      lo += half + 1;
      span -= half + 1;
    } else {
      span = half;
    }
  }
  // nothing found among all 128 values: the reference reads the value "at slot 128", i.e. whatever follows the
  // block in the caller's memory; here that read is not made and the value reported is 0, as in the batched form
  // (a 0-bit block has no memory behind its values: the answer is the base, as in the reference)
  *presult = lo < 128 || bit == 0 ? simdselectFOR(initvalue, in, bit, lo) : 0u;
  return lo;
}

// ---- single-block access to differentially coded blocks ----------------------
// One block in host memory, one answer: like the FOR accessors above these have no
// batch dimension (the batched forms are simdcomp_hip_select / _search / _scand1).
namespace {

// the field at `slot`, not yet summed
inline uint32_t field_at(const uint32_t *w, uint32_t bit, int slot) {
  const Field f = locate((size_t)slot, bit);
  uint32_t v = w[f.word] >> f.shift;
  if (f.split) v |= w[f.word + 4] << (32 - f.shift);
  return v & low_ones(bit);
}

// value number `slot` of the block: the offset plus the fields 0..slot (raw for 32 bits, the offset for 0 bits)
inline uint32_t d1_value(uint32_t init, const uint32_t *w, uint32_t bit, int slot) {
  if (bit == 0) return init;
  if (bit == 32) return w[slot];
  uint32_t v = init;
  for (int i = 0; i <= slot; ++i) v += field_at(w, bit, i);
  return v;
}

// src/simdpackedsearch.c:29
inline int bisect_raw(const uint32_t *a, uint32_t key, int imin, int imax) {
  imax--;
  while (imin + 1 < imax) {
    const int imid = imin + (imax - imin) / 2;
    if (a[imid] >= key) imax = imid;
    else imin = imid;
  }
  return a[imin] >= key ? imin : imax;
}

}  // namespace

// src/simdpackedselect.c:8096
uint32_t simdselectd1(uint32_t initvalue, const __m128i *in, uint32_t bit, int slot) {
  if (bit > 32) return (uint32_t)-1;
  return d1_value(initvalue, reinterpret_cast<const uint32_t *>(in), bit, slot & 127);
}

// src/simdpackedsearch.c:8284: first of the first `length` values that is >= key
int simdsearchwithlengthd1(uint32_t initvalue, const __m128i *in, uint32_t bit, int length, uint32_t key,
                           uint32_t *presult) {
  const uint32_t *w = reinterpret_cast<const uint32_t *>(in);
  if (bit > 32) return -1;
  if (bit == 32) {
    if (length > 0) {
      const int at = bisect_raw(w, key, 0, length);
      if (w[at] >= key) {
        *presult = w[at];
        return at;
      }
    }
  } else if (bit == 0) {
    if (length > 0 && initvalue >= key) {
      *presult = initvalue;
      return 0;
    }
  } else {
    uint32_t v = initvalue;
    for (int i = 0; i < length && i < 128; ++i) {
      v += field_at(w, bit, i);
      if (v >= key) {
        *presult = v;
        return i;
      }
    }
  }
  *presult = key + 1;
  return length;
}

// src/simdpackedsearch.c:16583: all 128 values; the offset vector ends up holding the group of four the search
// stopped in (the last group when nothing was found; untouched for 0 bits; the block's last vector for 32 bits,
// and only when the key was found).  Element 3 of the vector is the offset of the block.
int simdsearchd1(__m128i *initOffset, const __m128i *in, uint32_t bit, uint32_t key, uint32_t *presult) {
  const uint32_t *w = reinterpret_cast<const uint32_t *>(in);
  uint32_t *o = reinterpret_cast<uint32_t *>(initOffset);
  if (bit > 32) return -1;
  const uint32_t init = o[3];
  const int pos = simdsearchwithlengthd1(init, in, bit, 128, key, presult);
  if (bit == 0) return pos;
  if (bit == 32) {
    if (pos < 128)
      for (int i = 0; i < 4; ++i) o[i] = w[124 + i];
    return pos;
  }
  const int g = pos < 128 ? pos / 4 : 31;
  uint32_t v = d1_value(init, w, bit, 4 * g);
  o[0] = v;
  for (int i = 1; i < 4; ++i) o[i] = (v += field_at(w, bit, 4 * g + i));
  return pos;
}

// src/simdpackedselect.c:15216
void simdscand1(__m128i *initOffset, const __m128i *in, uint32_t bit) {
  const uint32_t *w = reinterpret_cast<const uint32_t *>(in);
  uint32_t *o = reinterpret_cast<uint32_t *>(initOffset);
  if (bit == 0 || bit > 32) return;
  const uint32_t init = o[3];
  for (int i = 0; i < 4; ++i) o[i] = d1_value(init, w, bit, 124 + i);
}

// src/simdbitpacking.c:14338: the b-bit window is cleared, then the raw value is
// ORed in (not masked, as in the reference); a split field replaces the low
// shift+b-32 bits of the lane's next word.
void simdfastset(__m128i *in128, uint32_t b, uint32_t value, size_t index) {
  uint32_t *w = reinterpret_cast<uint32_t *>(in128);
  if (b == 0) return;
  const Field f = locate(index, b);
  w[f.word] = (b == 32 ? 0u : (w[f.word] & ~(low_ones(b) << f.shift))) | (value << f.shift);
  if (f.split) {
    const uint32_t spill_bits = f.shift + b - 32;
    w[f.word + 4] = (w[f.word + 4] & ~low_ones(spill_bits)) | (value >> (32 - f.shift));
  }
}

// src/simdfor.c:15197
void simdfastsetFOR(uint32_t initvalue, __m128i *in, uint32_t bit, uint32_t value, size_t index) {
  simdfastset(in, bit, value - initvalue, index);
}

// src/simdintegratedbitpacking.c:25339
void simdfastsetd1fromprevious(__m128i *in, uint32_t bit, uint32_t previousvalue, uint32_t value, size_t index) {
  simdfastset(in, bit, value - previousvalue, index);
}

// src/simdintegratedbitpacking.c:25347
void simdfastsetd1(uint32_t initvalue, __m128i *in, uint32_t bit, uint32_t value, size_t index) {
  const uint32_t prev = index == 0 ? initvalue : simdselectd1(initvalue, in, bit, (int)index - 1);
  simdfastset(in, bit, value - prev, index);
}

}  // extern "C"
