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

// Select / search with the decode shared between neighbouring queries.  A wave takes RUN = 64 consecutive queries,
// lane j query j.  Queries that follow each other on the same block (a batch sorted by block, as the intersection of
// posting lists issues it; an unsorted batch simply has runs of length one) form one run: the block's width, offset
// and seed are fetched once (every lane fetches its own block's entries with one load per array and wave; runs are
// found by comparing with the lane before), the block is decoded once into LDS by one half-wave, and its queries are
// answered from there, every lane for itself, with one coalesced store per result array and wave at the end.
//   Measured on 2^24 searches over a corpus of 2^23 blocks (widths 1..20): see profiles/r03_microbench_query.json.
#ifndef SC_QUERY_RUN_U
#define SC_QUERY_RUN_U 4
#endif
constexpr int QUERY_RUN = 64;
constexpr int QUERY_RUN_U = SC_QUERY_RUN_U;  // blocks per half-wave whose payload loads are in flight together

template <int MODE, int WHAT, bool LAST4 = false, bool DIR = false>
__global__ __launch_bounds__(256) void k_query_runs(QueryArgs a) {
  __shared__ u32x4 lds_pay[4][2][33];
  // the decoded runs of one group of iterations, [2 * u + h]; 33 rows each, so that the runs start four banks apart:
  // every lane's bisection begins at value 64 of its own run, and with runs 128 words apart all of those were one bank
  __shared__ u32x4 lds_val[4][2 * QUERY_RUN_U][33];
  __shared__ uint32_t lds_head[4][QUERY_RUN + 1];  // lane of the first query of the wave's d-th run; [D] = 64
  constexpr int U = QUERY_RUN_U;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  u32x4 *slot = lds_pay[wv][h];
  const size_t q0 = ((size_t)blockIdx.x * 4 + wv) * QUERY_RUN;
  if (q0 >= a.nqueries) return;  // (whole waves; there is no workgroup barrier in this kernel)
  const size_t q = q0 + (size_t)lane;
  const bool valid = q < a.nqueries;
  const size_t qc = valid ? q : a.nqueries - 1;  // lanes past the end repeat the last query: they join its run
  const uintptr_t nothing = reinterpret_cast<uintptr_t>(&g_nothing);
  // first fetch: the query (block, key or slot, length), one coalesced load per array
  const uint64_t qb = load_global_as<uint64_t>(a.query_blocks ? reinterpret_cast<uintptr_t>(a.query_blocks + qc) : nothing);
  const void *ks = WHAT == Q_SELECT ? static_cast<const void *>(a.slots) : static_cast<const void *>(a.keys);
  const uint32_t my_key = load_global_as<uint32_t>(reinterpret_cast<uintptr_t>(ks) + 4 * qc);
  const int ln = load_global_as<int>(WHAT == Q_SEARCH && a.lengths ? reinterpret_cast<uintptr_t>(a.lengths + qc) : nothing);
  const int my_len = a.lengths ? ln : 128;
  const size_t blk = a.query_blocks ? (size_t)qb : qc;
  // second fetch: the block's width, offset and seed (lanes of one run read the same entries: one request), or its
  // directory entry, which holds all three
  uint8_t wb = 0;
  uint64_t po = 0;
  uint32_t sd = 0;
  if constexpr (DIR) {
    const u32x4 e = load_global_at(reinterpret_cast<uintptr_t>(a.directory) + DIRECTORY_ENTRY_BYTES * blk);
    wb = (uint8_t)(e.y >> 24);
    po = ((uint64_t)(e.y & 0xFFFFu) << 32) | e.x;
    sd = e.z;
  } else {
    wb = load_global_as<uint8_t>(a.bits ? reinterpret_cast<uintptr_t>(a.bits + blk) : nothing);
    po = load_global_as<uint64_t>(a.off ? reinterpret_cast<uintptr_t>(a.off + blk) : nothing);
    sd = load_global_as<uint32_t>(MODE != MODE_PLAIN && a.seeds ? reinterpret_cast<uintptr_t>(a.seeds + blk) : nothing);
  }
  // runs: a lane is a head when the lane before it asks for another block
  const uint32_t blk_lo = (uint32_t)blk, blk_hi = (uint32_t)((uint64_t)blk >> 32);
  const uint32_t prev_lo = (uint32_t)__shfl_up((int)blk_lo, 1, 64), prev_hi = (uint32_t)__shfl_up((int)blk_hi, 1, 64);
  const bool head = lane == 0 || prev_lo != blk_lo || prev_hi != blk_hi;
  const uint64_t heads = __ballot(head);
  const int D = __builtin_popcountll(heads);
  const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(heads >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)heads, 0u));
  if (head) lds_head[wv][below] = (uint32_t)lane;
  if (lane == 0) lds_head[wv][D] = QUERY_RUN;
  uint32_t my_b = (DIR || a.bits) ? wb : a.bit;
  my_b = my_b > 32 ? 32 : my_b;
  uint64_t my_base;
  uint32_t my_seed;
  if constexpr (DIR) {
    my_base = reinterpret_cast<uintptr_t>(a.payload) + po;
    my_seed = MODE == MODE_PLAIN ? 0u : sd;
  } else {
    const uint64_t has_off = a.off ? ~0ull : 0ull;
    my_base = reinterpret_cast<uintptr_t>(a.payload) + (po & has_off) + ((a.stream_layout ? blk + 1 : 0) & has_off) +
              ((blk * 16 * (size_t)my_b) & ~has_off);
    my_seed = MODE == MODE_PLAIN ? 0u : (a.seeds ? sd : a.seed0);
  }
  const uint32_t my_run = below + (head ? 1u : 0u) - 1u;  // the run this lane's query belongs to
  uint32_t res_value = 0;
  int res_pos = 0;
  u32x4 res_last4 = {0, 0, 0, 0};
  bool answered_above = false;
  wave_lds_fence();
  const int iters = (D + 1) >> 1;  // two runs at a time, one per half-wave
  for (int i0 = 0; i0 < iters; i0 += U) {
    uint32_t bs[U], seeds[U], first_lane[U], count[U];
    u32x4 lds_in[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int d = 2 * (i0 + u) + h;
      const bool have = d < D;
      const int dc = have ? d : 0;
      const uint32_t src = lds_head[wv][dc];
      const uint32_t nxt = lds_head[wv][dc + 1];
      first_lane[u] = src;
      count[u] = have ? nxt - src : 0u;
      const int from = 4 * (int)src;
      // (the crossbar reads only from lanes that are switched on: every lane takes part in every hand-over, the
      // half-wave without a run discards what it gets)
      const uint32_t head_b = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)my_b);
      bs[u] = have ? head_b : 0u;
      seeds[u] = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)my_seed);
      const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)(uint32_t)(my_base >> 32)) << 32) |
                            (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)(uint32_t)my_base);
      // every lane loads (lanes past the payload's end re-read its last vector, a zero-width block or an absent run
      // reads g_nothing): no load behind a branch
      const uint32_t last = bs[u] ? bs[u] - 1 : 0;
      const uintptr_t at = (uintptr_t)base + 16 * ((uint32_t)k < last ? (uint32_t)k : last);
      lds_in[u] = load_global_at(bs[u] ? at : nothing);
    }
    uint32_t slow_runs = 0;  // bit 2 * u + h: that run of the group is answered by its half-wave
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (2 * (i0 + u) >= D) break;  // wave-uniform
      const uint32_t b = bs[u];
      const uint32_t seed = seeds[u];
      u32x4 *vals4 = lds_val[wv][2 * u + h];
      const uint32_t *vals = reinterpret_cast<const uint32_t *>(vals4);
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
      // Blocks whose values wrap around 2^32 and raw 32-bit blocks cannot be bisected for "the first value >= key"
      // (a raw block is bisected in the reference's own way, src/simdpackedsearch.c:29): their half-wave answers
      // their queries here, one after the other, while the values are still in its registers.
      if constexpr (MODE == MODE_D1 && WHAT == Q_SEARCH) {
        const uint32_t before_w = dpp0<DPP_ROW_SHR1>(v.w);  // lanes at the start of a row of 16 get 0: fixed up
        const uint32_t prev_w = (k & 15) ? before_w : (k ? vals[4 * k - 1] : 0u);
        const uint64_t unordered = __ballot(b == 32 || v.x < prev_w || v.y < v.x || v.z < v.y || v.w < v.z);
        const bool slow0 = (uint32_t)unordered != 0, slow1 = (uint32_t)(unordered >> 32) != 0;
        slow_runs |= (slow0 ? 1u : 0u) << (2 * u) | (slow1 ? 2u : 0u) << (2 * u);
        const uint32_t c0 = slow0 ? (uint32_t)__builtin_amdgcn_readlane((int)count[u], 0) : 0u;
        const uint32_t c1 = slow1 ? (uint32_t)__builtin_amdgcn_readlane((int)count[u], 32) : 0u;
        const uint32_t rounds = c0 > c1 ? c0 : c1;  // both half-waves go round together
        const bool half_slow = h ? slow1 : slow0;
        for (uint32_t t = 0; t < rounds; ++t) {
          const bool mine = half_slow && t < count[u];
          const uint32_t ql = mine ? first_lane[u] + t : first_lane[u];  // the query's lane
          const uint32_t key = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (int)ql, (int)my_key);
          int len = __builtin_amdgcn_ds_bpermute(4 * (int)ql, my_len);
          len = len < 0 ? 0 : (len > 128 ? 128 : len);
          int spos;
          bool found;
          if (b == 32) {
            int imin = 0, imax = len - 1;
            while (imin + 1 < imax) {
              const int imid = imin + (imax - imin) / 2;
              if (vals[imid] >= key) imax = imid;
              else imin = imid;
            }
            spos = (len > 0 && vals[imin] >= key) ? imin : imax;
            found = len > 0 && vals[spos] >= key;
          } else {
            const uint32_t base4 = 4 * (uint32_t)k;
            const uint32_t m0 = half_ballot((int)base4 + 0 < len && v.x >= key, h);
            const uint32_t m1 = half_ballot((int)base4 + 1 < len && v.y >= key, h);
            const uint32_t m2 = half_ballot((int)base4 + 2 < len && v.z >= key, h);
            const uint32_t m3 = half_ballot((int)base4 + 3 < len && v.w >= key, h);
            spos = 128;
            if (m0) spos = min(spos, 4 * __builtin_ctz(m0) + 0);
            if (m1) spos = min(spos, 4 * __builtin_ctz(m1) + 1);
            if (m2) spos = min(spos, 4 * __builtin_ctz(m2) + 2);
            if (m3) spos = min(spos, 4 * __builtin_ctz(m3) + 3);
            found = spos < 128;
          }
          const uint32_t sval = found ? vals[spos & 127] : key + 1;
          spos = found ? spos : len;
          if constexpr (LAST4) {
            // what simdsearchd1 leaves in its offset vector (meaningful for length 128)
            u32x4 last = {seed, seed, seed, seed};
            if (b == 32) {
              if (found) last = vals4[31];
            } else if (b != 0) {
              last = vals4[found ? spos >> 2 : 31];
            }
            if (mine && k == 0 && q0 + ql < a.nqueries) *reinterpret_cast<u32x4_u *>(a.last4_out + 4 * (q0 + ql)) = last;
          }
          // (these few leave straight from here, one lane each; the lanes answered below store together at the end)
          if (mine && k == 0 && q0 + ql < a.nqueries) {
            a.positions_out[q0 + ql] = spos;
            a.values_out[q0 + ql] = sval;
          }
        }
      }
    }
    // ---- answers: once per group of U iterations, every lane for itself ------------------------------------------
    // A lane whose query belongs to one of the runs just decoded answers it from that run's values in LDS: select is
    // one read; the frame-of-reference search is the reference's own bisection (src/simdfor.c:15016), probe for
    // probe; the differential search (the first of the first `length` values that is >= key: the reference scans for
    // it, src/simdpackedsearch.c:8284) is the same bisection when the block's values do not decrease.
    {
      const bool my_turn = (int)(my_run / (2 * U)) == i0 / U;
      const uint32_t r = my_run % (2 * U);
      const bool my_slow = (slow_runs >> r) & 1u;
      const u32x4 *run4 = lds_val[wv][r];
      const uint32_t *run = reinterpret_cast<const uint32_t *>(run4);
      if constexpr (WHAT == Q_SELECT) {
        const uint32_t got = run[my_key & 127u];
        if (my_turn) res_value = got;
      } else {
        const int length = my_len < 0 ? 0 : (my_len > 128 ? 128 : my_len);
        // (for the differential search, whose values do not decrease here, a lower bound by fixed strides of 64, 32,
        // .. 1, 1 has half the instructions of this loop and measured the same: the probes' LDS latency is what counts)
        int begin = 0, cnt = length;
#pragma unroll
        for (int step_no = 0; step_no < 8; ++step_no) {  // 128 -> 0 in at most eight halvings
          const int step = cnt >> 1;
          const uint32_t probe = run[(begin + step) & 127];
          const bool go_up = cnt > 0 && probe < my_key;
          begin = go_up ? begin + step + 1 : begin;
          cnt = cnt > 0 ? (go_up ? cnt - step - 1 : step) : 0;
        }
        const uint32_t at_begin = run[begin & 127];
        int pos = begin;
        uint32_t value;
        if constexpr (MODE == MODE_D1) {
          const bool found = begin < length;
          value = found ? at_begin : my_key + 1;
          pos = found ? begin : length;
          if constexpr (LAST4) {
            // what simdsearchd1 leaves in its offset vector (meaningful for length 128)
            const u32x4 group = run4[found ? pos >> 2 : 31];
            const u32x4 mine4 = my_b == 0 ? u32x4{my_seed, my_seed, my_seed, my_seed} : group;
            if (my_turn && !my_slow) res_last4 = mine4;
          }
          if (my_turn && my_slow) answered_above = true;
        } else {
          value = begin < 128 ? at_begin : 0u;  // the reference reads past the block here
        }
        if (my_turn && !(MODE == MODE_D1 && my_slow)) {
          res_pos = pos;
          res_value = value;
        }
      }
    }
  }
  if (valid && !answered_above) {
    a.values_out[q] = res_value;
    if constexpr (WHAT == Q_SEARCH) {
      a.positions_out[q] = res_pos;
      if constexpr (LAST4) *reinterpret_cast<u32x4_u *>(a.last4_out + 4 * q) = res_last4;
    }
  }
}

#ifndef SC_QUERY_RUNS
#define SC_QUERY_RUNS 1
#endif

// One directory entry per block (layout: query_kernels.h).  An offset that does not fit 48 bits (no device memory is
// that large) becomes a zero-width block with seed 0, which is never dereferenced; word 3 marks it.
__global__ __launch_bounds__(256) void k_block_directory(uint32_t seed0, const uint32_t *seeds, const uint64_t *off,
                                                         const uint8_t *bits, size_t nblocks, uint32_t stream_layout,
                                                         u32x4 *directory) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= nblocks) return;
  uint32_t b = bits[k];
  b = b > 32 ? 32 : b;
  const uint64_t at = off[k] + (stream_layout ? k + 1 : 0);
  const uint32_t seed = seeds ? seeds[k] : seed0;
  const bool fits = at <= DIRECTORY_MAX_OFFSET;
  directory[k] = fits ? u32x4{(uint32_t)at, (uint32_t)(at >> 32) | (b << 24), seed, 0u} : u32x4{0u, 0u, 0u, 1u};
}

hipError_t launch_block_directory(uint32_t seed0, const uint32_t *seeds, const uint64_t *off, const uint8_t *bits, size_t nblocks,
                                  uint32_t stream_layout, void *directory, hipStream_t st) {
  if (nblocks == 0) return hipSuccess;
  hipLaunchKernelGGL(k_block_directory, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, st, seed0, seeds, off, bits, nblocks,
                     stream_layout, static_cast<u32x4 *>(directory));
  return hipGetLastError();
}

hipError_t launch_query(int mode, int what, const QueryArgs &a, hipStream_t st) {
  if (a.nqueries == 0) return hipSuccess;
  if (a.directory && what == Q_SCAN) return hipErrorInvalidValue;
  if ((SC_QUERY_RUNS || a.directory) && what != Q_SCAN) {
    const dim3 g((unsigned)((a.nqueries + 4 * QUERY_RUN - 1) / (4 * QUERY_RUN))), t(256);
#define SC_QR(MODE_, WHAT_, LAST4_)                                                          \
  do {                                                                                       \
    if (a.directory) hipLaunchKernelGGL((k_query_runs<MODE_, WHAT_, LAST4_, true>), g, t, 0, st, a);  \
    else hipLaunchKernelGGL((k_query_runs<MODE_, WHAT_, LAST4_, false>), g, t, 0, st, a);    \
  } while (0)
    if (mode == MODE_D1) {
      if (what == Q_SELECT) SC_QR(MODE_D1, Q_SELECT, false);
      else if (a.last4_out) SC_QR(MODE_D1, Q_SEARCH, true);
      else SC_QR(MODE_D1, Q_SEARCH, false);
    } else if (mode == MODE_FOR) {
      if (what == Q_SELECT) SC_QR(MODE_FOR, Q_SELECT, false);
      else SC_QR(MODE_FOR, Q_SEARCH, false);
    } else {
      if (what == Q_SELECT) SC_QR(MODE_PLAIN, Q_SELECT, false);
      else SC_QR(MODE_PLAIN, Q_SEARCH, false);
    }
#undef SC_QR
    return hipGetLastError();
  }
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
