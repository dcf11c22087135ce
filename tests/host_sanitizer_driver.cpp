// Test program for the host-only part of libsimdcomp_hip (csrc/scalar_access.cpp, csrc/host_arith.cpp), built with
// AddressSanitizer and UndefinedBehaviorSanitizer by tests/test_host_sanitizers.py and run on the CPU: every packed
// block lives in a heap buffer of exactly 16*b bytes, so a read or write one byte outside the block stops the run.
// Results are compared with the scalar restatement under oracle/ (compiled into this program with the same
// sanitizers).  Test infrastructure: nothing here is part of the library.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../include/simdcomp.h"
#include "../include/simdcomp_hip.h"

extern "C" {
uint32_t orc_bits(uint32_t v);
void orc_simdpackwithoutmask(const uint32_t *in, void *out, uint32_t bit);
void orc_simdunpack(const void *in, uint32_t *out, uint32_t bit);
void orc_simdpackwithoutmaskd1(uint32_t initvalue, const uint32_t *in, void *out, uint32_t bit);
void orc_simdunpackd1(uint32_t initvalue, const void *in, uint32_t *out, uint32_t bit);
void orc_simdpackFOR(uint32_t initvalue, const uint32_t *in, void *out, uint32_t bit);
void orc_simdunpackFOR(uint32_t initvalue, const void *in, uint32_t *out, uint32_t bit);
int64_t orc_compressedbytes(int64_t length, uint32_t bit);
uint32_t orc_simdselectFOR(uint32_t initvalue, const void *in, uint32_t bit, int slot);
int orc_simdsearchwithlengthFOR(uint32_t initvalue, const void *in, uint32_t bit, int length, uint32_t key, uint32_t *presult);
uint32_t orc_simdselectd1(uint32_t initvalue, const void *in, uint32_t bit, int slot);
int orc_simdsearchwithlengthd1(uint32_t initvalue, const void *in, uint32_t bit, int length, uint32_t key, uint32_t *presult);
int orc_simdsearchd1(uint32_t offset4[4], const void *in, uint32_t bit, uint32_t key, uint32_t *presult);
void orc_simdscand1(uint32_t offset4[4], const void *in, uint32_t bit);
void orc_simdfastset(void *in128, uint32_t b, uint32_t value, size_t index);
void orc_simdfastsetFOR(uint32_t initvalue, void *in, uint32_t bit, uint32_t value, size_t index);
}

static uint64_t g_state = 0x243F6A8885A308D3ull;
static uint64_t rnd() {
  uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static uint32_t mask_of(uint32_t b) { return b >= 32 ? 0xFFFFFFFFu : ((1u << b) - 1u); }

static long g_checks = 0;
#define CHECK(cond, ...)                                  \
  do {                                                    \
    ++g_checks;                                           \
    if (!(cond)) {                                        \
      fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
      fprintf(stderr, __VA_ARGS__);                       \
      fprintf(stderr, "\n");                              \
      exit(1);                                            \
    }                                                     \
  } while (0)

// a packed block in a heap allocation of exactly its size (b = 0: one byte that must never be touched)
struct Packed {
  uint8_t *p;
  size_t n;
  explicit Packed(uint32_t b) : n(16 * (size_t)b) { p = static_cast<uint8_t *>(aligned_alloc(16, n ? n : 16)); if (!n) { free(p); p = static_cast<uint8_t *>(malloc(1)); } }
  ~Packed() { free(p); }
  __m128i *v() { return reinterpret_cast<__m128i *>(p); }
};

static void accessors(uint32_t b) {
  uint32_t x[128], dec[128];
  for (int trial = 0; trial < 6; ++trial) {
    const uint32_t base = (uint32_t)rnd(), init = (uint32_t)rnd();
    // ---- frame of reference: values base + r, r < 2^b (wrapping) -----------------------------------------------
    for (int i = 0; i < 128; ++i) x[i] = base + ((uint32_t)rnd() & mask_of(b));
    if (trial & 1) {  // sorted offsets, so that the bisection has an order to work with
      uint32_t r[128];
      for (int i = 0; i < 128; ++i) r[i] = x[i] - base;
      for (int i = 1; i < 128; ++i) for (int j = i; j > 0 && r[j - 1] > r[j]; --j) { uint32_t t = r[j]; r[j] = r[j - 1]; r[j - 1] = t; }
      for (int i = 0; i < 128; ++i) x[i] = base + r[i];
    }
    Packed f(b);
    orc_simdpackFOR(base, x, f.p, b);
    for (int slot = 0; slot < 128; ++slot)
      CHECK(simdselectFOR(base, f.v(), b, slot) == orc_simdselectFOR(base, f.p, b, slot), "simdselectFOR b=%u slot=%d", b, slot);
    for (int q = 0; q < 40; ++q) {
      const int length = q < 8 ? 128 : (int)(rnd() % 129);
      uint32_t key = q == 0 ? 0u : q == 1 ? 0xFFFFFFFFu : x[rnd() % 128] + (uint32_t)(rnd() % 3) - 1u;
      uint32_t got = 7, want = 7;
      const int pg = simdsearchwithlengthFOR(base, f.v(), b, length, key, &got);
      // the restatement reads the value at slot 128 like the reference does; only call it when it stays inside
      int pw = 0, n = length;
      while (n > 0) { const int half = n / 2; if (orc_simdselectFOR(base, f.p, b, pw + half) < key) { pw += half + 1; n -= half + 1; } else n = half; }
      want = pw < 128 || b == 0 ? orc_simdselectFOR(base, f.p, b, pw) : 0u;
      CHECK(pg == pw && got == want, "simdsearchwithlengthFOR b=%u length=%d key=%u: %d/%u want %d/%u", b, length, key, pg, got, pw, want);
    }
    for (int q = 0; q < 24; ++q) {  // overwrite single values, compare the bytes and the decode
      const size_t index = q < 2 ? (q ? 127 : 0) : rnd() % 128;
      const uint32_t value = base + ((uint32_t)rnd() & mask_of(b));
      Packed g(b);
      memcpy(g.p, f.p, f.n);
      simdfastsetFOR(base, f.v(), b, value, index);
      orc_simdfastsetFOR(base, g.p, b, value, index);
      CHECK(memcmp(f.p, g.p, f.n) == 0, "simdfastsetFOR b=%u index=%zu", b, index);
      // (a 32-bit block holds raw values, src/simdfor.c:6954, yet simdfastsetFOR subtracts the base at every width,
      // src/simdfor.c:15197: what lands in the block then is value - base, in the reference too)
      if (b) x[index] = b == 32 ? value - base : value;
      orc_simdunpackFOR(base, f.p, dec, b);
      for (int i = 0; i < 128; ++i) CHECK(dec[i] == (b ? x[i] : base), "simdfastsetFOR decode b=%u", b);
      // plain simdfastset with a raw (unmasked) value
      const uint32_t raw = (uint32_t)rnd();
      memcpy(g.p, f.p, f.n);
      simdfastset(f.v(), b, raw, index);
      orc_simdfastset(g.p, b, raw, index);
      CHECK(memcmp(f.p, g.p, f.n) == 0, "simdfastset b=%u index=%zu", b, index);
      orc_simdunpackFOR(base, f.p, x, b);  // what the block holds now
    }
    // ---- differential: sorted values with gaps < 2^b (arbitrary 32-bit values for b = 32) ------------------------
    uint32_t v = init;
    for (int i = 0; i < 128; ++i) x[i] = b == 32 ? (uint32_t)rnd() : (v += (uint32_t)rnd() & mask_of(b));
    Packed d(b);
    orc_simdpackwithoutmaskd1(init, x, d.p, b);
    for (int slot = -3; slot < 140; ++slot)
      CHECK(simdselectd1(init, d.v(), b, slot) == orc_simdselectd1(init, d.p, b, slot), "simdselectd1 b=%u slot=%d", b, slot);
    for (int q = 0; q < 40; ++q) {
      const int length = q < 8 ? 128 : (int)(rnd() % 129);
      const uint32_t key = q == 0 ? 0u : q == 1 ? 0xFFFFFFFFu : x[rnd() % 128] + (uint32_t)(rnd() % 3) - 1u;
      uint32_t got = 7, want = 7;
      const int pg = simdsearchwithlengthd1(init, d.v(), b, length, key, &got);
      const int pw = orc_simdsearchwithlengthd1(init, d.p, b, length, key, &want);
      CHECK(pg == pw && got == want, "simdsearchwithlengthd1 b=%u length=%d key=%u: %d/%u want %d/%u", b, length, key, pg, got, pw, want);
      uint32_t o4[4] = {(uint32_t)rnd(), (uint32_t)rnd(), (uint32_t)rnd(), init}, w4[4];
      memcpy(w4, o4, 16);
      __m128i ov;
      memcpy(&ov, o4, 16);
      const int sg = simdsearchd1(&ov, d.v(), b, key, &got);
      const int sw = orc_simdsearchd1(w4, d.p, b, key, &want);
      memcpy(o4, &ov, 16);
      CHECK(sg == sw && got == want && memcmp(o4, w4, 16) == 0, "simdsearchd1 b=%u key=%u", b, key);
    }
    {
      uint32_t o4[4] = {1, 2, 3, init}, w4[4] = {1, 2, 3, init};
      __m128i ov;
      memcpy(&ov, o4, 16);
      simdscand1(&ov, d.v(), b);
      orc_simdscand1(w4, d.p, b);
      memcpy(o4, &ov, 16);
      CHECK(memcmp(o4, w4, 16) == 0, "simdscand1 b=%u", b);
    }
    if (b > 0 && b < 32) {  // simdfastsetd1: replace one value by another that keeps both neighbouring gaps in range
      for (int q = 0; q < 16; ++q) {
        const size_t index = q < 2 ? (q ? 127 : 0) : rnd() % 128;
        const uint32_t prev = index ? x[index - 1] : init;
        const uint32_t next = index < 127 ? x[index + 1] : prev + mask_of(b);
        const uint32_t room = next - prev, m = mask_of(b);  // both new gaps must fit b bits
        const uint32_t glo = room > m ? room - m : 0u, ghi = room < m ? room : m;
        const uint32_t value = prev + glo + (uint32_t)(rnd() % ((uint64_t)(ghi - glo) + 1));
        simdfastsetd1(init, d.v(), b, value, index);
        if (index < 127) simdfastsetd1fromprevious(d.v(), b, value, x[index + 1], index + 1);  // the next gap follows
        x[index] = value;
        orc_simdunpackd1(init, d.p, dec, b);
        for (int i = 0; i < 128; ++i) CHECK(dec[i] == x[i], "simdfastsetd1 b=%u index=%zu i=%d", b, index, i);
      }
    }
  }
}

static void sizes_and_shards() {
  for (uint32_t b = 0; b <= 32; ++b) {
    const int lens[] = {0, 1, 3, 4, 5, 127, 128, 129, 130, 9999, 1 << 20};
    for (int n : lens) {
      CHECK(simdpack_compressedbytes(n, b) == (int)orc_compressedbytes(n, b), "simdpack_compressedbytes(%d,%u)", n, b);
      CHECK(simdpackFOR_compressedbytes(n, b) == (int)orc_compressedbytes(n, b), "simdpackFOR_compressedbytes(%d,%u)", n, b);
    }
    const size_t big[] = {0, 1, 5, (size_t)1 << 30, ((size_t)1 << 33) + 5, (size_t)1 << 40, ((size_t)1 << 40) + 127};
    for (size_t n : big)
      CHECK(simdcomp_hip_compressedbytes(n, b) == (size_t)orc_compressedbytes((int64_t)n, b), "simdcomp_hip_compressedbytes(%zu,%u)", n, b);
  }
  for (uint32_t v : {0u, 1u, 2u, 255u, 256u, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFFu}) CHECK(bits(v) == orc_bits(v), "bits(%u)", v);
  size_t first = 0, count = 0;
  // equal block counts: more parts than blocks, nothing to share out, 2^40 blocks
  const size_t nbs[] = {0, 1, 3, 7, 1000003, (size_t)1 << 40, ((size_t)1 << 40) + 5};
  for (size_t nb : nbs)
    for (int parts : {1, 2, 3, 8, 64}) {
      size_t end = 0, lo = (size_t)-1, hi = 0;
      for (int p = 0; p < parts; ++p) {
        CHECK(simdcomp_hip_partition(nb, parts, p, &first, &count) == 0, "partition");
        CHECK(first == end, "partition not contiguous");
        end = first + count;
        lo = count < lo ? count : lo;
        hi = count > hi ? count : hi;
      }
      CHECK(end == nb && hi - lo <= 1, "partition nb=%zu parts=%d", nb, parts);
    }
  CHECK(simdcomp_hip_partition(10, 0, 0, &first, &count) == SIMDCOMP_HIP_ESIZE, "nparts 0");
  CHECK(simdcomp_hip_partition(10, 2, 2, &first, &count) == SIMDCOMP_HIP_ESIZE, "part out of range");
  CHECK(simdcomp_hip_partition(10, 2, -1, &first, &count) == SIMDCOMP_HIP_ESIZE, "negative part");
  CHECK(simdcomp_hip_partition(10, 2, 0, nullptr, &count) == SIMDCOMP_HIP_ESIZE, "NULL result");
  CHECK(strlen(simdcomp_hip_last_error()) > 0, "error text");
  // equal bytes: the offsets array has exactly nblocks + 1 entries (one read past it stops the run)
  for (size_t nb : {(size_t)0, (size_t)1, (size_t)2, (size_t)5, (size_t)1000, (size_t)77777}) {
    for (int shape = 0; shape < 4; ++shape) {
      std::vector<uint64_t> off(nb + 1);
      // shape 0: widths 1..20; 1: all zero-width; 2: one huge block in the middle; 3: offsets near 2^62 (a file
      // position inside a very large arena: only differences matter)
      uint64_t at = shape == 3 ? ((uint64_t)1 << 62) : 0;
      for (size_t k = 0; k < nb; ++k) {
        off[k] = at;
        uint64_t w = shape == 1 ? 0 : 1 + rnd() % 20;
        if (shape == 2 && k == nb / 2) w = (uint64_t)1 << 40;
        at += 16 * w;
      }
      off[nb] = at;
      for (int parts : {1, 2, 3, 8, 64}) {
        size_t end = 0;
        unsigned __int128 worst = 0;
        const unsigned __int128 whole = (unsigned __int128)(off[nb] - off[0]) + (unsigned __int128)nb * 512;
        for (int p = 0; p < parts; ++p) {
          CHECK(simdcomp_hip_partition_by_bytes(nb ? off.data() : nullptr, nb, parts, p, &first, &count) == 0, "partition_by_bytes");
          CHECK(first == end, "partition_by_bytes not contiguous nb=%zu parts=%d", nb, parts);
          end = first + count;
          CHECK(end <= nb, "partition_by_bytes beyond the corpus");
          const unsigned __int128 cost = (unsigned __int128)(off[first + count] - off[first]) + (unsigned __int128)count * 512;
          worst = cost > worst ? cost : worst;
        }
        CHECK(end == nb, "partition_by_bytes does not cover nb=%zu parts=%d", nb, parts);
        if (shape == 0 && nb >= 1000) {  // no shard is more than one block's cost above an equal share
          CHECK(worst <= whole / parts + 512 + 16 * 20 + 1, "partition_by_bytes unbalanced nb=%zu parts=%d", nb, parts);
        }
      }
    }
  }
  std::vector<uint64_t> two = {0, 16};
  CHECK(simdcomp_hip_partition_by_bytes(two.data(), 1, 0, 0, &first, &count) == SIMDCOMP_HIP_ESIZE, "nparts 0");
  CHECK(simdcomp_hip_partition_by_bytes(nullptr, 1, 1, 0, &first, &count) == SIMDCOMP_HIP_ESIZE, "NULL offsets");
  CHECK(simdcomp_hip_version() >= 100, "version");
}

int main() {
  for (uint32_t b = 0; b <= 32; ++b) accessors(b);
  sizes_and_shards();
  printf("host sanitizer driver: %ld checks passed\n", g_checks);
  return 0;
}
