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
  *presult = simdselectFOR(initvalue, in, bit, lo);
  return lo;
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

}  // extern "C"
