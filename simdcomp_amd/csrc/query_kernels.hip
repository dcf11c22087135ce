// Batched select / search / scan over packed blocks (gfx950 / CDNA4).
//
// One query per 32-lane half of a wave64.  The half-wave decodes its block the
// way the decoder does (payload vectors into a wave-private LDS slot, funnel
// shift, DPP prefix sum for d1), leaves the 128 values in a second LDS slot and
// answers from there, so a query costs one read of 16*b bytes and writes 4-24
// bytes.  The answers follow the reference routine by routine, including what
// it does on unsorted input:
//   simdselectd1            src/simdpackedselect.c:8096  (slot & 127, 0-bit block = offset)
//   simdsearchwithlengthd1  src/simdpackedsearch.c:8284  (widths 1..31 and 0: first of the first
//                           `length` values >= key, else length and key+1; width 32: the bisection of :29)
//   simdsearchd1, simdscand1  :16583, src/simdpackedselect.c:15216 (the four values left in the offset vector)
//   simdselectFOR, simdsearchwithlengthFOR  src/simdfor.c:14987, :15016 (lower-bound bisection, value at the
//                           returned slot even when nothing was found)
#include "query_kernels.h"

#include "codec_kernels.h"
#include "lane_ops.h"

namespace sc {

__device__ __forceinline__ uint32_t half_ballot(bool c, int h) {
  return (uint32_t)(__ballot(c) >> (32 * h));
}

// Queries per half-wave, their three rounds of fetches (block number; width / offset / seed; payload) issued together.
// Measured at 2^24 searches over random blocks: 5.9 / 7.8 / 8.7 G queries/s at 1 / 2 / 4.  (With the fetches behind
// per-query branches, as a first version had them, more than one per half-wave was slower.)
#ifndef SC_QUERY_U
#define SC_QUERY_U 4
#endif
constexpr int QUERY_U = SC_QUERY_U;

template <int MODE, int WHAT>
__global__ __launch_bounds__(256) void k_query(QueryArgs a) {
  __shared__ u32x4 lds_pay[4][2][33];
  __shared__ u32x4 lds_val[4][2][32];
  constexpr int U = QUERY_U;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  u32x4 *slot = lds_pay[wv][h];
  u32x4 *vals4 = lds_val[wv][h];
  const uint32_t *vals = reinterpret_cast<const uint32_t *>(vals4);
  const size_t first = ((size_t)blockIdx.x * 4 + wv) * 2 * U + h;
  size_t qcs[U], blks[U];
  uint32_t bs[U], seeds[U], keys_or_slots[U];
  int lens[U];
  const uint8_t *pbases[U];
  u32x4 lds_in[U];
  // Three rounds of loads, each round's loads side by side: (block id, key or slot, length) -> (width, offset,
  // seed) -> payload.  Optional arrays that are absent are "read" from g_nothing, so that no load hides behind a
  // branch of its own.
  const uintptr_t nothing = reinterpret_cast<uintptr_t>(&g_nothing);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t q = first + 2 * (size_t)u;
    qcs[u] = q < a.nqueries ? q : a.nqueries - 1;
    const uint64_t qb = load_global_as<uint64_t>(a.query_blocks ? reinterpret_cast<uintptr_t>(a.query_blocks + qcs[u]) : nothing);
    blks[u] = a.query_blocks ? (size_t)qb : qcs[u];
    const void *ks = WHAT == Q_SELECT ? static_cast<const void *>(a.slots) : static_cast<const void *>(a.keys);
    keys_or_slots[u] = load_global_as<uint32_t>(WHAT != Q_SCAN ? reinterpret_cast<uintptr_t>(ks) + 4 * qcs[u] : nothing);
    const int ln = load_global_as<int>(WHAT == Q_SEARCH && a.lengths ? reinterpret_cast<uintptr_t>(a.lengths + qcs[u]) : nothing);
    lens[u] = a.lengths ? ln : 128;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint8_t wb = load_global_as<uint8_t>(a.bits ? reinterpret_cast<uintptr_t>(a.bits + blks[u]) : nothing);
    const uint64_t po = load_global_as<uint64_t>(a.off ? reinterpret_cast<uintptr_t>(a.off + blks[u]) : nothing);
    const uint32_t sd = load_global_as<uint32_t>(MODE != MODE_PLAIN && a.seeds ? reinterpret_cast<uintptr_t>(a.seeds + blks[u]) : nothing);
    bs[u] = a.bits ? wb : a.bit;
    bs[u] = bs[u] > 32 ? 32 : bs[u];
    pbases[u] = static_cast<const uint8_t *>(a.payload);
    // both terms are always computed (masks instead of a choice): a choice would put the offset load behind a branch
    // that waits for the width
    const uint64_t has_off = a.off ? ~0ull : 0ull;
    pbases[u] += (po & has_off) + ((a.stream_layout ? blks[u] + 1 : 0) & has_off) + ((blks[u] * 16 * (size_t)bs[u]) & ~has_off);
    seeds[u] = MODE == MODE_PLAIN ? 0u : (a.seeds ? sd : a.seed0);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    // every lane loads (lanes past the payload's end re-read its last vector, a zero-width block reads g_nothing):
    // a predicated load would pull the offset load above under the same predicate, behind the wait for the width
    const uint32_t last = bs[u] ? bs[u] - 1 : 0;
    const uintptr_t at = reinterpret_cast<uintptr_t>(pbases[u]) + 16 * ((uint32_t)k < last ? (uint32_t)k : last);
    lds_in[u] = load_global_at(bs[u] ? at : nothing);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
  const size_t q = first + 2 * (size_t)u;
  const bool valid = q < a.nqueries;
  const uint32_t b = bs[u];
  const uint32_t seed = seeds[u];
  wave_lds_fence();
  slot[k] = lds_in[u];
  wave_lds_fence();
  const uint32_t p = (uint32_t)k * b, w = p >> 5, s = p & 31;
  const u32x4 lo = slot[w], hi = slot[w + 1];
  const uint32_t m = field_mask(b);
  u32x4 v;
  v.x = funnel(hi.x, lo.x, s) & m;
  v.y = funnel(hi.y, lo.y, s) & m;
  v.z = funnel(hi.z, lo.z, s) & m;
  v.w = funnel(hi.w, lo.w, s) & m;
  if (b < 32) {
    if constexpr (MODE == MODE_D1) {
      v.y += v.x;
      v.z += v.y;
      v.w += v.z;
      const uint32_t before = seed + half_scan_add(v.w) - v.w;
      v += before;
    } else {
      v += seed;
    }
  }
  wave_lds_fence();
  vals4[k] = v;
  wave_lds_fence();

  if constexpr (WHAT == Q_SELECT) {
    const uint32_t at = keys_or_slots[u] & 127u;
    if (valid && k == 0) a.values_out[q] = vals[at];
  } else if constexpr (WHAT == Q_SCAN) {
    if (valid && k == 31) *reinterpret_cast<u32x4_u *>(a.last4_out + 4 * q) = v;
  } else {
    const uint32_t key = keys_or_slots[u];
    int length = lens[u];
    length = length < 0 ? 0 : (length > 128 ? 128 : length);
    int pos;
    uint32_t value;
    if constexpr (MODE == MODE_D1) {
      bool found;
      if (b == 32) {
        // raw values, bisected exactly as src/simdpackedsearch.c:29 does
        int imin = 0, imax = length - 1;
        while (imin + 1 < imax) {
          const int imid = imin + (imax - imin) / 2;
          if (vals[imid] >= key) imax = imid;
          else imin = imid;
        }
        pos = (length > 0 && vals[imin] >= key) ? imin : imax;
        found = length > 0 && vals[pos] >= key;
      } else {
        const uint32_t base = 4 * (uint32_t)k;
        const uint32_t m0 = half_ballot((int)base + 0 < length && v.x >= key, h);
        const uint32_t m1 = half_ballot((int)base + 1 < length && v.y >= key, h);
        const uint32_t m2 = half_ballot((int)base + 2 < length && v.z >= key, h);
        const uint32_t m3 = half_ballot((int)base + 3 < length && v.w >= key, h);
        pos = 128;
        if (m0) pos = min(pos, 4 * __builtin_ctz(m0) + 0);
        if (m1) pos = min(pos, 4 * __builtin_ctz(m1) + 1);
        if (m2) pos = min(pos, 4 * __builtin_ctz(m2) + 2);
        if (m3) pos = min(pos, 4 * __builtin_ctz(m3) + 3);
        found = pos < 128;
      }
      value = found ? vals[pos & 127] : key + 1;
      pos = found ? pos : length;
      if (a.last4_out) {
        // what simdsearchd1 leaves in its offset vector (meaningful for length 128)
        u32x4 last = {seed, seed, seed, seed};
        if (b == 32) {
          if (found) last = vals4[31];
        } else if (b != 0) {
          last = vals4[found ? pos >> 2 : 31];
        }
        if (valid && k == 0) *reinterpret_cast<u32x4_u *>(a.last4_out + 4 * q) = last;
      }
    } else {
      int begin = 0, count = length;
      while (count > 0) {
        const int step = count / 2;
        if (vals[begin + step] < key) {
          begin += step + 1;
          count -= step + 1;
        } else {
          count = step;
        }
      }
      pos = begin;
      value = begin < 128 ? vals[begin] : 0u;  // the reference reads past the block here
    }
    if (valid && k == 0) {
      a.positions_out[q] = pos;
      a.values_out[q] = value;
    }
  }
  }
}

hipError_t launch_query(int mode, int what, const QueryArgs &a, hipStream_t st) {
  if (a.nqueries == 0) return hipSuccess;
  const dim3 g((unsigned)((a.nqueries + 8 * QUERY_U - 1) / (8 * QUERY_U))), t(256);
#define SC_Q(MODE_, WHAT_) hipLaunchKernelGGL((k_query<MODE_, WHAT_>), g, t, 0, st, a)
  if (mode == MODE_D1) {
    if (what == Q_SELECT) SC_Q(MODE_D1, Q_SELECT);
    else if (what == Q_SEARCH) SC_Q(MODE_D1, Q_SEARCH);
    else SC_Q(MODE_D1, Q_SCAN);
  } else if (mode == MODE_FOR) {
    if (what == Q_SELECT) SC_Q(MODE_FOR, Q_SELECT);
    else if (what == Q_SEARCH) SC_Q(MODE_FOR, Q_SEARCH);
    else return hipErrorInvalidValue;
  } else {
    if (what == Q_SELECT) SC_Q(MODE_PLAIN, Q_SELECT);
    else if (what == Q_SEARCH) SC_Q(MODE_PLAIN, Q_SEARCH);
    else return hipErrorInvalidValue;
  }
#undef SC_Q
  return hipGetLastError();
}

}  // namespace sc
