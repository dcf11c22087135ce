// Device kernels for the 128-integer block bit-packing codec (gfx950 / CDNA4).
//
// Wire format (identical to lemire/simdcomp, include/simdcomputil.h:22 and the
// schedule visible at src/simdbitpacking.c:217-323): value i of a block lives in
// lane i&3, row k = i>>2; a lane is a bit stream in which row k occupies bits
// [k*b, k*b+b); word w of lane l is stored at uint32 index 4*w+l, so "vector" w
// (16 bytes) holds word w of all four lanes.  A block is b vectors.
//
// Thread mapping used by every kernel here: one block per 32-lane half of a
// wave64, thread k of the half owns row k (the four values 4k..4k+3, one from
// each lane, one 16-byte access on the integer side).  A 256-thread workgroup
// therefore covers 8 blocks per step and runs U steps per loop iteration with
// all of their global loads issued up front.
//
//   unpack: the half-wave's b payload vectors are fetched by its first b
//           threads with one coalesced 16-byte load each, staged in a
//           wave-private LDS slot, and every thread then reads the one or two
//           vectors that hold its row (word extraction through LDS); a funnel
//           shift (v_alignbit_b32) + mask yields the four values.  No
//           workgroup barrier: a wave only ever reads what it wrote itself.
//   d1:     running sum = 3 adds inside the thread + a 32-lane inclusive scan
//           of the per-thread totals done with DPP row_shr:1/2/4/8 and
//           row_bcast:15 restricted to rows 1 and 3 (so nothing leaks from one
//           half-wave's block into the other's), seeded by the block offset.
//   pack:   every thread shifts its row into place, receives the spill of the
//           previous row through DPP wave_shr:1, and a segmented OR-scan over
//           the threads that share an output word (again DPP row_shr +
//           row_bcast:15) leaves the finished word in the last thread of each
//           segment, which stores it.  No LDS, no barrier.
//
// All arithmetic is uint32 mod 2^32.  Memory-bound: no MFMA anywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "codec_kernels.h"
#include "lane_ops.h"

namespace sc {

constexpr int BLOCKS_PER_STEP = 8;  // 256 threads = 8 half-waves

// ---------------------------------------------------------------------------
// unpack
// ---------------------------------------------------------------------------
// B >= 0: width fixed at compile time (uniform batches); B < 0: per-block width
// from a.bits (or the uniform a.bit), decided per half-wave at run time.
// TAIL: the last block may hold fewer than 128 values (a.length counts values);
// only rows/values that exist are read and written.
template <int B, int MODE, int U, bool TAIL>
__global__ __launch_bounds__(256) void k_unpack_flex(UnpackArgs a) {
  __shared__ u32x4 lds[4][2][33];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  u32x4 *slot = lds[wv][h];
  const uint8_t *payload_bytes = static_cast<const uint8_t *>(a.payload);
  // 4-lane groups per wide block (see UnpackArgs): 1, 2 or 4, as a shift
  const uint32_t ls = a.lane_groups > 2 ? 2u : (a.lane_groups > 1 ? 1u : 0u);
  const size_t S = (size_t)1 << ls;
  const size_t per_iter = (size_t)BLOCKS_PER_STEP * U;
  const size_t ntiles = (a.nblocks + per_iter - 1) / per_iter;
  const uintptr_t nothing = reinterpret_cast<uintptr_t>(&g_nothing);

  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t first = tile * per_iter + (size_t)wv * 2 * U + h;
    size_t blk[U];
    bool valid[U];
    uint32_t b[U], seed[U];
    const uint8_t *pbase[U];
    u32x4 ld[U];
    // Every load is issued whatever the block (blocks past the end repeat the last one, absent arrays are "read"
    // from g_nothing, rows past the payload's end re-read its last vector; only the stores are predicated): a load
    // behind a branch of its own would wait for all the loads in front of it.
#pragma unroll
    for (int u = 0; u < U; ++u) {
      blk[u] = first + 2 * (size_t)u;
      valid[u] = blk[u] < a.nblocks;
      const size_t bc = valid[u] ? blk[u] : a.nblocks - 1;
      const uint32_t wb = load_global_as<uint8_t>(B < 0 && a.bits ? reinterpret_cast<uintptr_t>(a.bits + bc) : nothing);
      const uint64_t po = load_global_as<uint64_t>(a.off ? reinterpret_cast<uintptr_t>(a.off + bc) : nothing);
      const uint32_t sd = load_global_as<uint32_t>(MODE != MODE_PLAIN && a.seeds ? reinterpret_cast<uintptr_t>(a.seeds + bc) : nothing);
      if constexpr (B >= 0) {
        b[u] = B;
      } else {
        b[u] = a.bits ? (valid[u] ? (wb > 32 ? 32u : wb) : 0u) : a.bit;  // a width byte above 32 counts as 32
      }
      const uint64_t has_off = a.off ? ~0ull : 0ull;
      const uint64_t by_off = po + (a.stream_layout ? a.block_base + bc + 1 : 0);
      const uint64_t uniform = (bc >> ls) * (16 * (size_t)b[u] * S) + 16 * (bc & (S - 1));
      pbase[u] = payload_bytes + ((by_off & has_off) | (uniform & ~has_off));
      if constexpr (MODE != MODE_PLAIN) seed[u] = a.seeds ? sd : a.seed0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint32_t nvec = b[u];
      if constexpr (TAIL) {
        const size_t start = blk[u] * 128;
        const size_t nval = start < a.length ? (a.length - start < 128 ? a.length - start : 128) : 0;
        nvec = (uint32_t)((((nval + 3) / 4) * b[u] + 31) / 32);
      }
      const uint32_t row = (uint32_t)k < nvec ? (uint32_t)k : (nvec ? nvec - 1 : 0);
      ld[u] = load_global_at(nvec ? reinterpret_cast<uintptr_t>(pbase[u] + 16 * S * row) : nothing);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      u32x4 v;
      if constexpr (B == 0) {
        v = u32x4{0, 0, 0, 0};
      } else if constexpr (B == 32) {
        v = ld[u];
      } else {
        wave_lds_fence();
        if ((uint32_t)k < b[u]) slot[k] = ld[u];
        wave_lds_fence();
        const uint32_t p = (uint32_t)k * b[u], w = p >> 5, s = p & 31;
        const u32x4 lo = slot[w], hi = slot[w + 1];
        const uint32_t m = field_mask(b[u]);
        v.x = funnel(hi.x, lo.x, s) & m;
        v.y = funnel(hi.y, lo.y, s) & m;
        v.z = funnel(hi.z, lo.z, s) & m;
        v.w = funnel(hi.w, lo.w, s) & m;
      }
      if constexpr (MODE == MODE_D1) {
        // running sum over the block, seeded by the block offset; a 32-bit wide
        // block holds raw values and is left alone
        u32x4 r = v;
        r.y += r.x;
        r.z += r.y;
        r.w += r.z;
        const uint32_t incl = half_scan_add(r.w);
        const uint32_t before = seed[u] + incl - r.w;
        r.x += before;
        r.y += before;
        r.z += before;
        r.w += before;
        if (B == 32 || (B < 0 && b[u] == 32)) r = v;
        v = r;
      } else if constexpr (MODE == MODE_FOR) {
        if (!(B == 32 || (B < 0 && b[u] == 32))) {
          v.x += seed[u];
          v.y += seed[u];
          v.z += seed[u];
          v.w += seed[u];
        }
      }
      if (valid[u]) {
        uint32_t *o = a.out + (blk[u] >> ls) * (128 * S) + 4 * S * k + 4 * (blk[u] & (S - 1));
        if constexpr (TAIL) {
          const size_t idx = blk[u] * 128 + 4 * (size_t)k;
          if (idx + 4 <= a.length) {
            *reinterpret_cast<u32x4 *>(o) = v;
          } else {
            if (idx + 0 < a.length) o[0] = v.x;
            if (idx + 1 < a.length) o[1] = v.y;
            if (idx + 2 < a.length) o[2] = v.z;
          }
        } else {
          *reinterpret_cast<u32x4 *>(o) = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// pack
// ---------------------------------------------------------------------------
constexpr int max_rows_per_word(int b) { return b <= 0 ? 32 : (32 + b - 1) / b; }

// GENERIC reproduces the reference's arbitrary-length packer
// (src/simdbitpacking.c:14190 simdpack_shortlength, src/simdfor.c:15043
// simdpackFOR_length): values past a.length are padding (0, or the frame base),
// a field that ends exactly on a word boundary still hands value >> b to the
// next word (visible only for out-of-range input), that hand-over also crosses
// from one block to the next, and only the vectors that hold bits are written.
template <int B, int MODE, bool MASKED, int U, bool GENERIC>
__global__ __launch_bounds__(256) void k_pack_flex(PackArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  uint8_t *payload_bytes = static_cast<uint8_t *>(a.payload);
  const size_t S = a.lane_groups > 1 ? a.lane_groups : 1;  // 4-lane groups per wide block (see UnpackArgs)
  const size_t per_iter = (size_t)BLOCKS_PER_STEP * U;
  const size_t ntiles = (a.nblocks + per_iter - 1) / per_iter;

  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t first = tile * per_iter + (size_t)wv * 2 * U + h;
    size_t blk[U];
    bool valid[U];
    uint32_t b[U], seed[U];
    uint8_t *pbase[U];
    u32x4 x[U], carry[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      blk[u] = first + 2 * (size_t)u;
      valid[u] = blk[u] < a.nblocks;
      if constexpr (B >= 0) {
        b[u] = B;
      } else {
        b[u] = a.bits ? (valid[u] ? a.bits[blk[u]] : 0u) : a.bit;
        b[u] = b[u] > 32 ? 32 : b[u];  // a width byte above 32 counts as 32
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      pbase[u] = payload_bytes;
      if (a.off) {
        if (valid[u]) pbase[u] += a.off[blk[u]] + (a.stream_layout ? a.chain_base + blk[u] + 1 : 0);
      } else {
        pbase[u] += (blk[u] / S) * (16 * (size_t)b[u] * S) + 16 * (blk[u] % S);
      }
      seed[u] = a.seed0;
      if constexpr (MODE == MODE_D1) {
        // explicit per-block offsets, or the chained stream of example.c:63-71:
        // block 0 starts from seed0, block k from the last value of block k-1
        if (a.seeds) {
          if (valid[u]) seed[u] = a.seeds[blk[u]];
        } else if (valid[u] && (blk[u] > 0 || a.chain_base > 0)) {
          seed[u] = a.in[blk[u] * 128 - 1];
        }
      } else if constexpr (MODE == MODE_FOR) {
        if (a.seeds && valid[u]) seed[u] = a.seeds[blk[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t idx = (blk[u] / S) * (128 * S) + 4 * S * (size_t)k + 4 * (blk[u] % S);
      x[u] = u32x4{0, 0, 0, 0};
      carry[u] = u32x4{0, 0, 0, 0};
      if constexpr (GENERIC) {
        const uint32_t pad = (MODE == MODE_FOR) ? seed[u] : 0u;
        x[u] = u32x4{pad, pad, pad, pad};
        if (valid[u]) {
          if (idx + 4 <= a.length) {
            x[u] = *reinterpret_cast<const u32x4 *>(a.in + idx);
          } else {
            if (idx + 0 < a.length) x[u].x = a.in[idx + 0];
            if (idx + 1 < a.length) x[u].y = a.in[idx + 1];
            if (idx + 2 < a.length) x[u].z = a.in[idx + 2];
          }
          // last row of the previous block: its fields end on the block's final
          // word boundary and hand value >> b to this block's first word
          if (k == 0 && (blk[u] > 0 || a.carry_first) && a.carry_blocks) carry[u] = *reinterpret_cast<const u32x4 *>(a.in + idx - 4);
        }
      } else {
        if (valid[u]) x[u] = *reinterpret_cast<const u32x4 *>(a.in + idx);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t bu = b[u];
      const bool raw = (B == 32) || (B < 0 && bu == 32);
      u32x4 v = x[u];
      if constexpr (MODE == MODE_D1) {
        uint32_t prev = dpp0<DPP_WAVE_SHR1>(v.w);
        if (k == 0) prev = seed[u];
        v = u32x4{x[u].x - prev, x[u].y - x[u].x, x[u].z - x[u].y, x[u].w - x[u].z};
      } else if constexpr (MODE == MODE_FOR) {
        v -= seed[u];
      }
      if constexpr (MASKED) v &= field_mask(bu);

      const uint32_t p = (uint32_t)k * bu, w = p >> 5, s = p & 31;
      u32x4 acc = v << s;
      // what this row hands to the next word
      const bool spills = GENERIC ? (s + bu >= 32) : (s + bu > 32);
      const uint32_t back = (32 - s) & 31;
      u32x4 sp = spills ? (v >> back) : u32x4{0, 0, 0, 0};
      u32x4 from_prev;
      from_prev.x = dpp0<DPP_WAVE_SHR1>(sp.x);
      from_prev.y = dpp0<DPP_WAVE_SHR1>(sp.y);
      from_prev.z = dpp0<DPP_WAVE_SHR1>(sp.z);
      from_prev.w = dpp0<DPP_WAVE_SHR1>(sp.w);
      if (k == 0) {
        from_prev = u32x4{0, 0, 0, 0};
        if constexpr (GENERIC) {
          if ((blk[u] > 0 || a.carry_first) && a.carry_blocks && bu > 0 && bu < 32) {
            u32x4 c = carry[u];
            if constexpr (MODE == MODE_FOR) c -= seed[u];
            from_prev = c >> bu;
          }
        }
      }
      acc |= from_prev;

      // OR together the threads whose rows start in the same word
      const uint32_t w15 = (15u * bu) >> 5;
#define SC_SEG_STEP(D, CTRL)                                                             \
  if (B < 0 || D < max_rows_per_word(B)) {                                               \
    /* masked at the giving lane (row k + D starts in the same word iff D * b <= its bit offset): see k_pack_fast */ \
    const uint32_t give = ((uint32_t)D * bu <= ((((uint32_t)k + D) * bu) & 31u)) ? 0xFFFFFFFFu : 0u; \
    acc.x |= dpp0<CTRL>(acc.x & give);                                                   \
    acc.y |= dpp0<CTRL>(acc.y & give);                                                   \
    acc.z |= dpp0<CTRL>(acc.z & give);                                                   \
    acc.w |= dpp0<CTRL>(acc.w & give);                                                   \
  }
      SC_SEG_STEP(1, DPP_ROW_SHR1)
      SC_SEG_STEP(2, DPP_ROW_SHR2)
      SC_SEG_STEP(4, DPP_ROW_SHR4)
      SC_SEG_STEP(8, DPP_ROW_SHR8)
#undef SC_SEG_STEP
      {
        const uint32_t same = (k >= 16 && w15 == w) ? 0xFFFFFFFFu : 0u;
        acc.x |= dpp0<DPP_ROW_BCAST15, 0xA>(acc.x) & same;
        acc.y |= dpp0<DPP_ROW_BCAST15, 0xA>(acc.y) & same;
        acc.z |= dpp0<DPP_ROW_BCAST15, 0xA>(acc.z) & same;
        acc.w |= dpp0<DPP_ROW_BCAST15, 0xA>(acc.w) & same;
      }
      const bool last = (k == 31) || ((((uint32_t)(k + 1) * bu) >> 5) != w);
      uint32_t nvec = bu;
      if constexpr (GENERIC) {
        const size_t start = blk[u] * 128;
        const size_t nval = start < a.length ? (a.length - start < 128 ? a.length - start : 128) : 0;
        nvec = (uint32_t)((((nval + 3) / 4) * bu + 31) / 32);
      }
      if (valid[u]) {
        if (raw) {
          if (!GENERIC) *reinterpret_cast<u32x4_u *>(pbase[u] + 16 * S * k) = x[u];
        } else if (last && w < nvec) {
          *reinterpret_cast<u32x4_u *>(pbase[u] + 16 * S * w) = acc;
        }
        if (!GENERIC && a.off && a.stream_layout && k == 0) pbase[u][-1] = (uint8_t)bu;
        if (MODE == MODE_D1 && a.seeds_out && k == 0) a.seeds_out[blk[u]] = seed[u];
      }
    }
  }
}


// ---------------------------------------------------------------------------
// fast paths: one-shot workgroups, straight-line bodies
// ---------------------------------------------------------------------------
// Two measured facts about gfx950 shape these kernels (tools/exp/*.hip are the
// probes that produced them):
//
//  1. Loads and stores of a wave retire through ONE in-order counter (vmcnt).
//     Behind a divergent branch the compiler can no longer count what is in
//     flight and waits for everything (s_waitcnt vmcnt(0)) before each LDS
//     write and each store.  The bodies below have no branch: every block a
//     workgroup touches exists, every array it indexes is present, lanes past
//     the end of a payload re-read its last vector instead of being masked.
//     Ragged ends and optional-argument forms go to the flex kernels.
//
//  2. HBM streams fastest when the set of addresses in flight across the chip
//     is a small window that advances through memory.  A grid of many short
//     workgroups (each handles 8*U blocks = 4*U KiB of integers and ends) gives
//     exactly that: the dispatcher hands out block ranges in order as slots
//     free up.  A persistent grid-stride loop with the same body keeps tens of
//     MiB in flight at scattered offsets and measured 15-25 % slower; so did
//     larger per-workgroup footprints.  U is therefore small (1 or 2).
//
//   B in 0..32: uniform width, payload of block k at k*16*B; B == -1: per-block
//   widths a.bits[] and payload offsets a.off[].  a.seeds[] must be an array
//   for the d1 / FOR modes.  Launch with grid = nblocks / (8*U) workgroups.

//  3. With so little in flight per wave, a wave's lifetime is what limits the
//     narrow widths: its payload read waits on HBM behind the chip's store
//     traffic before the stores can even be issued.  When PF > 0 every PF decode
//     workgroups are preceded in the grid by one workgroup that only touches
//     the payload (and seeds) of the group `dist` groups further on, so the
//     decode waves find their input in the memory-side cache.  The prefetching
//     waves do wait on HBM, but they are 1 in PF + 1.  (Issuing the same
//     prefetch from the decode waves themselves gains nothing: a wave keeps
//     its slot until its last load has returned.)

// Decode one already-loaded block per half-wave: LDS exchange, extraction, the
// mode's arithmetic, one 16-byte store per thread.  `b` is B when B >= 0.
template <int B, int MODE>
__device__ __forceinline__ u32x4 unpack_values(const u32x4 ld, const uint32_t b, const uint32_t seed, u32x4 *slot, const int k) {
  u32x4 v;
  if constexpr (B == 0) {
    v = u32x4{0, 0, 0, 0};
  } else if constexpr (B == 32) {
    v = ld;
  } else {
    wave_lds_fence();
    slot[k] = ld;
    wave_lds_fence();
    const uint32_t p = (uint32_t)k * b, w = p >> 5, s = p & 31;
    const u32x4 lo = slot[w], hi = slot[w + 1];
    const uint32_t m = field_mask(b);
    v.x = funnel(hi.x, lo.x, s) & m;
    v.y = funnel(hi.y, lo.y, s) & m;
    v.z = funnel(hi.z, lo.z, s) & m;
    v.w = funnel(hi.w, lo.w, s) & m;
  }
  if constexpr (MODE == MODE_D1 && B != 32) {
    u32x4 r = v;
    r.y += r.x;
    r.z += r.y;
    r.w += r.z;
    const uint32_t incl = half_scan_add(r.w);
    const uint32_t before = seed + incl - r.w;
    r.x += before;
    r.y += before;
    r.z += before;
    r.w += before;
    if constexpr (B < 0) {
      const bool raw = b == 32;
      v.x = raw ? v.x : r.x;
      v.y = raw ? v.y : r.y;
      v.z = raw ? v.z : r.z;
      v.w = raw ? v.w : r.w;
    } else {
      v = r;
    }
  } else if constexpr (MODE == MODE_FOR && B != 32) {
    const uint32_t add = (B < 0 && b == 32) ? 0u : seed;
    v.x += add;
    v.y += add;
    v.z += add;
    v.w += add;
  }
  return v;
}

template <int B, int MODE, int STORE = 0>
__device__ __forceinline__ void unpack_finish(const u32x4 ld, const uint32_t b, const uint32_t seed, u32x4 *slot,
                                              const int k, uint32_t *out_row) {
  const u32x4 v = unpack_values<B, MODE>(ld, b, seed, slot, k);
  if constexpr (STORE == 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(out_row));
  else if constexpr (STORE == 2) *reinterpret_cast<u32x4 *>(out_row) = v;
  else stream_store(reinterpret_cast<u32x4 *>(out_row), v);
}

// Touch the 16-byte vectors that hold bytes [byte0, byte0 + NBYTES) of `base` with ordinary (cache-allocating)
// loads, all issued back to back; vectors past the one holding `last_byte` re-read that one.  `base` needs no
// alignment: addresses are rounded down to 16, which stays inside the 16-byte chunk the array starts in.
template <int BS, size_t NBYTES>
__device__ __forceinline__ void touch_bytes(const void *base, size_t byte0, size_t last_byte, u32x4 &acc) {
  constexpr int NV = (int)(NBYTES / 16) + 1;  // + 1: an unaligned range straddles one more vector
  constexpr int TRIPS = (NV + BS - 1) / BS;
  const uintptr_t v0 = ((uintptr_t)base + byte0) >> 4, vlast = ((uintptr_t)base + last_byte) >> 4;
#pragma unroll
  for (int t = 0; t < TRIPS; ++t) {
    uintptr_t v = v0 + (uintptr_t)t * BS + threadIdx.x;
    v = v < vlast ? v : vlast;
    acc |= load_global_at(v << 4);
  }
}

// LS > 0: the wide layouts (2^LS four-lane groups per block, see UnpackArgs::lane_groups): group g of wide block W is
// "block" W * 2^LS + g here, its payload vectors and output rows 2^LS apart.  Uniform widths, plain coding only.
// From 20 bits on the payload is read with non-temporal loads: it is 40-50 % of the traffic there and read
// once, and keeping it out of the caches' way measured +2..+4 points of HBM peak for b = 26..32 and, together with
// 256 threads per workgroup, +1 point and steadier rates for b = 20..25 (neutral below 20).
#ifndef SC_NT_PAYLOAD_MINB
#define SC_NT_PAYLOAD_MINB 20
#endif
// (not for the 16-lane layout, whose payload vectors of one block lie 64 bytes apart: 71-72 -> 73 % at 24 and 32 bits)
#ifndef SC_NT_PAYLOAD_MAXLS
#define SC_NT_PAYLOAD_MAXLS 2
#endif
#ifndef SC_NT_MIXED_UNPACK
#define SC_NT_MIXED_UNPACK 0
#endif
// per-block widths: the wave's metadata through the scalar unit (2-5 % faster from 9 bits on and on the byte stream,
// neutral for 5..8 bits; workgroup shapes re-swept with it: 256 x 2 and 128 x 2 decode blocks of 12..24 bits 3-7 %
// faster but narrow ones 15-25 % slower, 512 x 4 gains 1 % below 10 bits and loses 4 % at 24: 256 x 4 stays)
#ifndef SC_MIXED_SCALAR_META
#define SC_MIXED_SCALAR_META 1
#endif
// per-block widths: the prefetch workgroups also touch a group's payload when its blocks average at most this many
// bits (0: never), at DIST4 quarters of the metadata's distance.  Measured with 2^30 integers: groups of 3 / 4 / 5 / 6
// bits decode 17 / 15 / 11 / 4 % faster, 7 bits and more gain nothing and a corpus of random widths averaging 7.2
// bits loses 4 % (so the limit is 6); the payload must be fetched 30-40 MiB of output ahead -- at 5 MiB the same
// prefetch costs 3-6 %, at 10 MiB it changes nothing, at 20 MiB it gains two thirds of what it gains at 30.
#ifndef SC_PF_MIXED_PAYLOAD_MAXB
#define SC_PF_MIXED_PAYLOAD_MAXB 6
#endif
#ifndef SC_PF_MIXED_PAYLOAD_DIST4
#define SC_PF_MIXED_PAYLOAD_DIST4 2
#endif
template <int B, int MODE, int U, int BS, int PF = 0, int LS = 0>
__global__ __launch_bounds__(BS) void k_unpack_fast(UnpackArgs a, uint32_t nwg = 0, uint32_t dist = 0) {
  static_assert(LS == 0 || (B >= 0 && MODE == MODE_PLAIN), "wide layouts: uniform width, plain coding");
  __shared__ u32x4 lds[BS / 64][2][33];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  u32x4 *slot = lds[wv][h];
  const u32x4 *payload = reinterpret_cast<const u32x4 *>(a.payload);
  constexpr size_t per_iter = (size_t)(BS / 32) * U;

  size_t tile = blockIdx.x;
  if constexpr (PF > 0) {
    static_assert(B != 0, "nothing to prefetch for zero-width blocks");
    const uint32_t group = blockIdx.x / (PF + 1), r = blockIdx.x - group * (PF + 1);
    if (r == 0) {
      // groups [t0, t1): two per prefetcher while the distance ramps up from the start of the array, one after
      const size_t t0 = (size_t)group + (group + 1 < dist ? group + 1 : dist);
      const size_t t1 = (size_t)group + 1 + (group + 2 < dist ? group + 2 : dist);
      constexpr size_t GB = (size_t)PF * per_iter;
      const size_t nblk = (size_t)nwg * per_iter;
      u32x4 acc = {0, 0, 0, 0};
      for (size_t t = t0; t < t1; ++t) {
        if (t * GB >= nblk) break;
        if constexpr (B > 0) {
          touch_bytes<BS, GB * B * 16>(a.payload, t * GB * B * 16, nblk * B * 16 - 1, acc);
          // (no seed array: the same bytes of the payload are touched instead, which keeps this path free of branches)
          if constexpr (MODE != MODE_PLAIN)
            touch_bytes<BS, GB * 4>(a.seeds ? static_cast<const void *>(a.seeds) : a.payload, t * GB * 4, nblk * 4 - 1, acc);
        } else {
          // per-block widths: the metadata (widths, payload offsets, seeds), so that the decode waves' first
          // round of loads hits the cache; where the payload lies is in that metadata (below)
          touch_bytes<BS, GB>(a.bits, t * GB, nblk - 1, acc);
          touch_bytes<BS, GB * 8>(a.off, t * GB * 8, nblk * 8 - 1, acc);
          if constexpr (MODE != MODE_PLAIN) touch_bytes<BS, GB * 4>(a.seeds, t * GB * 4, nblk * 4 - 1, acc);
        }
      }
      if constexpr (B < 0 && SC_PF_MIXED_PAYLOAD_MAXB > 0) {
        // ... and the payload of the group(s) a fraction of that distance ahead, whose metadata an earlier prefetcher
        // has brought in: the bytes from the first block's payload to the end of the last block's, if that is a span
        // a group of narrow blocks laid end to end could fill (payloads in any other order are left alone)
        const size_t hp = (size_t)dist * SC_PF_MIXED_PAYLOAD_DIST4 / 4;
        const size_t p0 = (size_t)group + (group + 1 < hp ? group + 1 : hp);
        const size_t p1 = (size_t)group + 1 + (group + 2 < hp ? group + 2 : hp);
        for (size_t p = p0; p < p1 && hp > 0; ++p) {
          const size_t fb = p * GB;
          if (fb >= nblk) break;
          const size_t lb = (fb + GB < nblk ? fb + GB : nblk) - 1;
          const uint32_t wl = a.bits[lb];
          uint64_t lo = a.off[fb], hi = a.off[lb] + 16 * (uint64_t)(wl > 32 ? 32 : wl);
          if (a.stream_layout) {
            lo += a.block_base + fb + 1;
            hi += a.block_base + lb + 1;
          }
          const uint64_t span = hi - lo;  // (wraps to a huge value when the last block lies in front of the first)
          if (span - 1 < (uint64_t)(lb - fb + 1) * (16 * SC_PF_MIXED_PAYLOAD_MAXB + 1)) {
            // a fixed number of loads, all in flight together; those past the end re-read the last vector
            constexpr int TRIPS = (int)((GB * (16 * SC_PF_MIXED_PAYLOAD_MAXB + 1) / 16 + BS) / BS);
            const uintptr_t v0 = (reinterpret_cast<uintptr_t>(a.payload) + lo) >> 4;
            const uintptr_t vlast = (reinterpret_cast<uintptr_t>(a.payload) + hi - 1) >> 4;
#pragma unroll
            for (int trip = 0; trip < TRIPS; ++trip) {
              uintptr_t v = v0 + (uintptr_t)trip * BS + threadIdx.x;
              v = v < vlast ? v : vlast;
              acc |= load_global_at(v << 4);
            }
          }
        }
      }
      asm volatile("" ::"v"(acc));  // keeps the loads; the only wait of this workgroup
      return;
    }
    tile = (size_t)group * PF + (r - 1);
    if (tile >= nwg) return;
  }
  {
    const size_t first = tile * per_iter + (size_t)wv * 2 * U + h;
    uint32_t b[U], seed[U];
    const u32x4_u *src[U];  // uniform widths: always 16-byte aligned; per-block widths: any alignment in stream layout
    u32x4 ld[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t blk = first + 2 * (size_t)u;
      if constexpr (B >= 0) {
        b[u] = B;
        const size_t row = (size_t)(k < B ? k : (B ? B - 1 : 0));
        src[u] = reinterpret_cast<const u32x4_u *>(payload + (((blk >> LS) * (size_t)B) << LS) + (blk & ((1u << LS) - 1)) + (row << LS));
      }
    }
    if constexpr (B < 0) {
      // width, offset and seed of the wave's 2 * U consecutive blocks: one coalesced load per array, side by side
      // (12 memory instructions fewer per wave than per-lane loads: 9 points of HBM peak).  The payload address
      // depends on them: two memory latencies, not three
      uint64_t o[U];
      constexpr int NB = 2 * U;
#if SC_MIXED_SCALAR_META
      // ... through the scalar unit: the three addresses are the same for the whole wave, the values arrive in scalar
      // registers, and the wave issues three vector memory instructions fewer
      typedef const __attribute__((address_space(4))) uint32_t *k32;
      typedef const __attribute__((address_space(4))) uint64_t *k64;
      const size_t wbase = tile * per_iter + (size_t)__builtin_amdgcn_readfirstlane(wv) * NB;
      const k64 offp = (k64)(uintptr_t)(a.off + wbase);
      const uintptr_t wat = (uintptr_t)(a.bits + wbase);  // 2 * U width bytes at any alignment: the dwords around them
      const k32 wp = (k32)(wat & ~(uintptr_t)3);
      const uint32_t sh = (uint32_t)(wat & 3) * 8;
      static_assert(NB == 8 || NB == 4 || NB == 2, "widths of a wave's blocks: at most 8 bytes");
      const uint32_t w0 = wp[0], w1 = NB > 4 || sh ? wp[NB > 4 || sh ? 1 : 0] : 0;
      const uint32_t w2 = NB > 4 ? wp[sh ? 2 : 0] : 0;
      const uint64_t wlo = ((uint64_t)w1 << 32) | w0;
      const uint64_t widths = sh ? (wlo >> sh) | ((uint64_t)w2 << (64 - sh)) : wlo;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t b0 = (uint32_t)(widths >> (16 * u)) & 0xFF, b1 = (uint32_t)(widths >> (16 * u + 8)) & 0xFF;
        const uint32_t bb = h ? b1 : b0;
        b[u] = bb > 32 ? 32 : bb;  // a width byte above 32 is treated as 32 (nothing is indexed with it)
        const uint64_t o0 = offp[2 * u], o1 = offp[2 * u + 1];
        o[u] = h ? o1 : o0;
        if constexpr (MODE != MODE_PLAIN) {
          const k32 sp = (k32)(uintptr_t)(a.seeds + wbase);
          const uint32_t s0 = sp[2 * u], s1 = sp[2 * u + 1];
          seed[u] = h ? s1 : s0;
        }
      }
#else
      const size_t mine = tile * per_iter + (size_t)wv * NB + (size_t)(lane % NB);
      const uint32_t mb_raw = a.bits[mine];
      const uint64_t mo = a.off[mine];
      uint32_t ms = 0;
      if constexpr (MODE != MODE_PLAIN) ms = a.seeds[mine];
      const uint32_t mb = mb_raw > 32 ? 32 : mb_raw;  // a width byte above 32 is treated as 32 (nothing is indexed with it)
      hand_out<U>(mb, h, b);
      hand_out<U>(mo, h, o);
      if constexpr (MODE != MODE_PLAIN) hand_out<U>(ms, h, seed);
#endif
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t blk = first + 2 * (size_t)u;
        const uint32_t last = b[u] ? b[u] - 1 : 0;
        // stream layout: the record's width byte and the k width bytes before it sit in front of the payload
        const uint64_t skip = a.stream_layout ? a.block_base + blk + 1 : 0;
        uintptr_t at = reinterpret_cast<uintptr_t>(a.payload) + o[u] + skip + 16 * ((uint32_t)k < last ? (uint32_t)k : last);
        // the lanes past the payload's end only keep the access pattern branch-free: they read the 16-byte line that
        // holds the start of its last vector (32 lanes on one unaligned address coalesce worse than on an aligned one)
        at = (uint32_t)k > last ? (at & ~(uintptr_t)15) : at;
        const u32x4_u *in_payload = reinterpret_cast<const u32x4_u *>(at);
        // a zero-width block has no payload: it reads (and ignores) 16 spare bytes instead (reading its own output
        // area, as this kernel once did, is 512 bytes of HBM traffic per such block: 15 % on a corpus where half are)
        src[u] = b[u] ? in_payload : reinterpret_cast<const u32x4_u *>(&g_nothing);
      }
    }
    if constexpr (B != 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr ((B >= SC_NT_PAYLOAD_MINB && LS < SC_NT_PAYLOAD_MAXLS) || (B < 0 && SC_NT_MIXED_UNPACK)) ld[u] = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) u32x4_u *>(reinterpret_cast<uintptr_t>(src[u])));
        else ld[u] = load_global_at(reinterpret_cast<uintptr_t>(src[u]));
      }
    }
    if constexpr (B >= 0 && MODE != MODE_PLAIN) {
      // The seeds of the wave's 2 * U consecutive blocks: one coalesced load, handed out through scalar registers.
      // It goes out behind the payload loads and without a branch of its own (no array -- one offset / base for every
      // block, as in the *_length forms -- reads g_nothing instead): a wait for it in front of the payload loads
      // would put two memory latencies in a row (measured: 80 -> 60-70 % of peak for widths 15..31).
      // (The same load through the scalar unit, as the per-block-width kernel fetches its metadata: within +-0.5 points
      // for 1..7 bits, -2..3 points for 8..16 bits, nothing from 17 on.)
      const size_t mine = tile * per_iter + (size_t)wv * 2 * U + (size_t)(lane % (2 * U));
      const uint32_t ms = load_global_as<uint32_t>(a.seeds ? reinterpret_cast<uintptr_t>(a.seeds + mine)
                                                           : reinterpret_cast<uintptr_t>(&g_nothing));
      hand_out<U>(ms, h, seed);
#pragma unroll
      for (int u = 0; u < U; ++u) seed[u] = a.seeds ? seed[u] : a.seed0;
    }
    if constexpr (LS == 2) {
      // 16-lane layout: the wave's four blocks (U = 2) are the four lane groups of ONE wide block, whose 2 KiB of
      // output interleave them 16 bytes at a time.  Stored from where they are decoded, an instruction would write 32
      // bytes of every 64; the values go through LDS once more instead (padded rows: no bank conflicts), and every
      // store instruction writes 1 KiB in one piece.
      static_assert(U == 2, "16-lane layout: two steps per half-wave");
      __shared__ u32x4 xch[BS / 64][4][33];
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        v[u] = unpack_values<B, MODE>(B != 0 ? ld[u] : u32x4{0, 0, 0, 0}, b[u], 0u, slot, k);
      wave_lds_fence();
#pragma unroll
      for (int u = 0; u < U; ++u) xch[wv][2 * u + h][k] = v[u];
      wave_lds_fence();
      uint32_t *wide = a.out + (tile * per_iter + (size_t)wv * 4) * 128;  // four blocks of 128 = one wide block of 512
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int idx = j * 64 + lane;  // 16-byte piece of the wide block: row idx / 4, lane group idx % 4
        stream_store(reinterpret_cast<u32x4 *>(wide + 4 * idx), xch[wv][idx & 3][idx >> 2]);
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t blk = first + 2 * (size_t)u;
        unpack_finish<B, MODE>(B != 0 ? ld[u] : u32x4{0, 0, 0, 0}, b[u], MODE != MODE_PLAIN ? seed[u] : 0u, slot, k,
                               a.out + (((blk >> LS) * 128) << LS) + 4 * (blk & ((1u << LS) - 1)) + ((4 * (size_t)k) << LS));
      }
    }
  }
}

// GENERIC: full blocks of the reference's arbitrary-length frame-of-reference packer (src/simdfor.c:15043
// simdpackFOR_length): a field that ends exactly on a word boundary still hands value >> b to the next word, across
// block boundaries too (visible only for out-of-range input).  The partial last block stays with k_pack_flex.
#ifndef SC_NT_PACK_MINB
#define SC_NT_PACK_MINB 1
#endif
#ifndef SC_NT_PACKIN_MINB
#define SC_NT_PACKIN_MINB 1
#endif
// The 16-lane layout (LS = 2) reads every 64-byte line of its input with two instructions (32 bytes each): a
// non-temporal first read would make the second one fetch the line again (63-71 % of HBM peak against 73-78 % with
// ordinary loads); its payload stores stay non-temporal up to 8 bits only (b = 16..32: 64-70 % against 72-76 %).
#ifndef SC_NT_PACK_LS2_MAXB
#define SC_NT_PACK_LS2_MAXB 8
#endif
#ifndef SC_NT_MIXED
#define SC_NT_MIXED 1
#endif
template <int B, int MODE, bool MASKED, int U, int BS, int LS = 0, bool GENERIC = false>
__global__ __launch_bounds__(BS) void k_pack_fast(PackArgs a) {
  static_assert(LS == 0 || (B >= 0 && MODE == MODE_PLAIN), "wide layouts: uniform width, plain coding");
  static_assert(!GENERIC || (B > 0 && B < 32 && MODE == MODE_FOR && !MASKED && LS == 0), "generic: FOR, widths 1..31");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  u32x4 *payload = reinterpret_cast<u32x4 *>(a.payload);
  constexpr size_t per_iter = (size_t)(BS / 32) * U;

  {
    const size_t tile = blockIdx.x;
    const size_t first = tile * per_iter + (size_t)wv * 2 * U + h;
    uint32_t b[U], seed[U];
    uint8_t *pbase[U];
    u32x4 x[U], acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t blk = first + 2 * (size_t)u;
      {
        const u32x4 *row = reinterpret_cast<const u32x4 *>(a.in + (((blk >> LS) * 128) << LS) + 4 * (blk & ((1u << LS) - 1)) + ((4 * (size_t)k) << LS));
        if constexpr ((B >= SC_NT_PACKIN_MINB && LS < 2) || (B < 0 && SC_NT_MIXED)) x[u] = __builtin_nontemporal_load(row);
        else x[u] = stream_load(row);
      }
      if constexpr (B >= 0) {
        b[u] = B;
        pbase[u] = reinterpret_cast<uint8_t *>(payload + (((blk >> LS) * (size_t)B) << LS) + (blk & ((1u << LS) - 1)));
      }
    }
    // Per-block metadata, every lane loading its own block's entry: the encoders are bound by vector-ALU work, and
    // handing coalesced loads out through scalar registers (as the decoders do) costs them 3-6 points of HBM peak;
    // scalar loads of widths and offsets (as the per-block-width decoder has them) cost 4 %.
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t blk = first + 2 * (size_t)u;
      if constexpr (B < 0) {
        b[u] = a.bits[blk];
        pbase[u] = static_cast<uint8_t *>(a.payload) + a.off[blk] + (a.stream_layout ? a.chain_base + blk + 1 : 0);
      }
      if constexpr (MODE == MODE_D1) {
        // explicit seeds, or chained from the input: block k starts from in[128k-1], block 0 from seed0
        // (address chosen by arithmetic, value chosen after the load: no branch in front of it)
        const bool start = blk == 0 && a.chain_base == 0;  // first block of the whole stream
        const uintptr_t at = a.seeds ? reinterpret_cast<uintptr_t>(a.seeds + blk)
                                     : reinterpret_cast<uintptr_t>(a.in + (start ? 0 : blk * 128 - 1));
        seed[u] = load_global_as<uint32_t>(at);
      } else if constexpr (MODE == MODE_FOR) {
        // (no array: one base for every block, as in simdpackFOR_length; read from g_nothing to stay branch-free)
        seed[u] = load_global_as<uint32_t>(a.seeds ? reinterpret_cast<uintptr_t>(a.seeds + blk)
                                                   : reinterpret_cast<uintptr_t>(&g_nothing));
      }
    }
    // Nothing moves across this line, and nothing above it uses a loaded value: otherwise the scheduler issues the
    // seed loads of the later steps only after it has waited for the first step's integers (one in-order counter),
    // which puts a second memory latency behind the first.
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (B < 0) b[u] = b[u] > 32 ? 32 : b[u];  // as in the decoder: a width byte above 32 counts as 32
      if constexpr (MODE == MODE_D1) {
        const bool start = first + 2 * (size_t)u == 0 && a.chain_base == 0;
        seed[u] = (!a.seeds && start) ? a.seed0 : seed[u];
      } else if constexpr (MODE == MODE_FOR) {
        seed[u] = a.seeds ? seed[u] : a.seed0;
      }
    }
    // GENERIC: what the last row of the block in front hands to this block's first word.  For the wave's first block
    // it comes from memory, for the others from the neighbouring half-wave (below).
    u32x4 handed = {0, 0, 0, 0};
    if constexpr (GENERIC) {
      const size_t wb = tile * per_iter + (size_t)wv * 2 * U;
      const bool has_prev = wb > 0 || a.carry_first;
      const u32x4 last_row = *reinterpret_cast<const u32x4 *>(a.in + (has_prev ? wb * 128 - 4 : 0));
      // (masked, not chosen: a choice would move the load under the condition, behind a wait of its own)
      handed = ((last_row - seed[0]) >> (uint32_t)B) & (has_prev ? 0xFFFFFFFFu : 0u);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t bu = b[u];
      u32x4 v = x[u];
      if constexpr (MODE == MODE_D1) {
        uint32_t prev = dpp0<DPP_WAVE_SHR1>(v.w);
        prev = k == 0 ? seed[u] : prev;
        v = u32x4{x[u].x - prev, x[u].y - x[u].x, x[u].z - x[u].y, x[u].w - x[u].z};
      } else if constexpr (MODE == MODE_FOR) {
        v -= seed[u];
      }
      if constexpr (MASKED) v &= field_mask(bu);
      const uint32_t p = (uint32_t)k * bu, w = p >> 5, s = p & 31;
      u32x4 t = v << s;
      const uint32_t back = (32 - s) & 31;
      const uint32_t keep = (GENERIC ? (s + bu >= 32) : (s + bu > 32)) ? 0xFFFFFFFFu : 0u;
      u32x4 sp = (v >> back) & keep;
      if constexpr (GENERIC) {
        // lane 32 takes lane 31's hand-over as it comes (the block in front is the other half-wave's); lane 0 takes
        // what lane 63 handed over at the step before, or `handed` at the first step
        u32x4 fp = {dpp0<DPP_WAVE_SHR1>(sp.x), dpp0<DPP_WAVE_SHR1>(sp.y), dpp0<DPP_WAVE_SHR1>(sp.z), dpp0<DPP_WAVE_SHR1>(sp.w)};
        fp = lane == 0 ? handed : fp;
        t |= fp;
        handed = u32x4{(uint32_t)__builtin_amdgcn_readlane((int)sp.x, 63), (uint32_t)__builtin_amdgcn_readlane((int)sp.y, 63),
                       (uint32_t)__builtin_amdgcn_readlane((int)sp.z, 63), (uint32_t)__builtin_amdgcn_readlane((int)sp.w, 63)};
      } else {
        // (row 0 takes nothing: lane 0 has no lane in front of it, and lane 32's neighbour, row 31 of the other
        // block, ends exactly on a word boundary -- 32 * b bits -- so its spill is zero by `keep`)
        t.x |= dpp0<DPP_WAVE_SHR1>(sp.x);
        t.y |= dpp0<DPP_WAVE_SHR1>(sp.y);
        t.z |= dpp0<DPP_WAVE_SHR1>(sp.z);
        t.w |= dpp0<DPP_WAVE_SHR1>(sp.w);
      }
#define SC_SEG_STEP(D, CTRL)                                                          \
  if constexpr (B < 0 || D < max_rows_per_word(B)) {                                  \
    /* Lane j hands its partial word to lane j + D when row j + D starts in the same word, i.e. D * b <= bit */ \
    /* offset of row j + D.  Masking at the GIVING lane leaves "t |= shifted" for the taker, which the       */ \
    /* compiler folds into one v_or_b32 with a DPP operand (a mask at the taker costs two more instructions    */ \
    /* per component: the AND cannot take the DPP operand, its neutral element is not the zero fill).          */ \
    const uint32_t give = ((uint32_t)D * bu <= ((((uint32_t)k + D) * bu) & 31u)) ? 0xFFFFFFFFu : 0u;            \
    t.x |= dpp0<CTRL>(t.x & give);                                                     \
    t.y |= dpp0<CTRL>(t.y & give);                                                     \
    t.z |= dpp0<CTRL>(t.z & give);                                                     \
    t.w |= dpp0<CTRL>(t.w & give);                                                     \
  }
      SC_SEG_STEP(1, DPP_ROW_SHR1)
      if constexpr (B < 0) {
        // per-block widths: a word holds at most ceil(32 / b) rows, so the wider both blocks of this wave are, the
        // fewer doubling steps their segments need (wave-uniform branches; a skipped step would have added nothing)
        if (__any(bu < 16)) {
          SC_SEG_STEP(2, DPP_ROW_SHR2)
          if (__any(bu < 8)) {
            SC_SEG_STEP(4, DPP_ROW_SHR4)
            if (__any(bu < 4)) { SC_SEG_STEP(8, DPP_ROW_SHR8) }
          }
        }
      } else {
        SC_SEG_STEP(2, DPP_ROW_SHR2)
        SC_SEG_STEP(4, DPP_ROW_SHR4)
        SC_SEG_STEP(8, DPP_ROW_SHR8)
      }
#undef SC_SEG_STEP
      if constexpr (B < 0 || ((15 * B) >> 5) == ((16 * B) >> 5)) {
        const uint32_t same = (k >= 16 && ((15u * bu) >> 5) == w) ? 0xFFFFFFFFu : 0u;
        t.x |= dpp0<DPP_ROW_BCAST15, 0xA>(t.x) & same;
        t.y |= dpp0<DPP_ROW_BCAST15, 0xA>(t.y) & same;
        t.z |= dpp0<DPP_ROW_BCAST15, 0xA>(t.z) & same;
        t.w |= dpp0<DPP_ROW_BCAST15, 0xA>(t.w) & same;
      }
      acc[u] = t;
    }
    // stores last, so that everything above is one straight line
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t bu = b[u];
      const uint32_t w = ((uint32_t)k * bu) >> 5;
      if constexpr (B == 32) {
        if constexpr (B >= SC_NT_PACK_MINB && (LS < 2 || B <= SC_NT_PACK_LS2_MAXB)) __builtin_nontemporal_store(x[u], reinterpret_cast<u32x4 *>(pbase[u] + ((16 * k) << LS)));
        else *reinterpret_cast<u32x4 *>(pbase[u] + ((16 * k) << LS)) = x[u];
      } else if constexpr (B > 0) {
        const bool last = (k == 31) || ((((uint32_t)(k + 1) * bu) >> 5) != w);
        if constexpr (B >= SC_NT_PACK_MINB && (LS < 2 || B <= SC_NT_PACK_LS2_MAXB)) {
          if (last) __builtin_nontemporal_store(acc[u], reinterpret_cast<u32x4 *>(pbase[u] + ((16 * w) << LS)));
        } else {
          if (last) *reinterpret_cast<u32x4 *>(pbase[u] + ((16 * w) << LS)) = acc[u];
        }
      } else if constexpr (B < 0) {
        const bool raw = bu == 32;
        const bool last = (k == 31) || ((((uint32_t)(k + 1) * bu) >> 5) != w);
        const u32x4 val = raw ? x[u] : acc[u];
        if (bu != 0 && (raw || last)) {  // raw: w == k
          if constexpr (SC_NT_MIXED) __builtin_nontemporal_store(val, reinterpret_cast<u32x4_u *>(pbase[u] + 16 * w));
          else *reinterpret_cast<u32x4_u *>(pbase[u] + 16 * w) = val;
        }
        // (the width byte leaves the way the payload next to it does: 0.908 -> 0.894 ms at 2^30 integers)
        if (a.stream_layout && k == 0) {
          if constexpr (SC_NT_MIXED) __builtin_nontemporal_store((uint8_t)bu, pbase[u] - 1);
          else pbase[u][-1] = (uint8_t)bu;
        }
      }
    }
    if constexpr (MODE == MODE_D1) {
      if (a.seeds_out && k == 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) a.seeds_out[first + 2 * (size_t)u] = seed[u];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// per-block reductions: maxbits / maxbitsd1 / min-max (+ frame width)
// ---------------------------------------------------------------------------
#ifndef SC_NT_REDUCE
#define SC_NT_REDUCE 1
#endif
__device__ __forceinline__ u32x4 reduce_load(const u32x4 *p) {
#if SC_NT_REDUCE
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

template <int WHAT, int U>
__global__ __launch_bounds__(256) void k_reduce(ReduceArgs a) {
  __shared__ uint32_t minmax_rows[WHAT == RED_MINMAX ? 2 : 1][WHAT == RED_MINMAX ? 4 : 1][2][WHAT == RED_MINMAX ? U : 1][32];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  const size_t per_iter = (size_t)BLOCKS_PER_STEP * U;
  const size_t ntiles = (a.nblocks + per_iter - 1) / per_iter;
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t first = tile * per_iter + (size_t)wv * 2 * U + h;
    u32x4 x[U];
    uint32_t seed[U];
    bool valid[U];
    uint32_t out_bits = 0, out_seed = 0;
    // every load is issued unconditionally (a block past the end re-reads the last one; only the stores are
    // predicated), and the U loads of a kind back to back: behind a per-step branch they would wait for one another
    size_t bc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t blk = first + 2 * (size_t)u;
      valid[u] = blk < a.nblocks;
      bc[u] = valid[u] ? blk : a.nblocks - 1;
      seed[u] = a.seed0;
    }
    if (WHAT == RED_MAXBITS && a.group > 1) {
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = u32x4{0, 0, 0, 0};
      for (uint32_t g = 0; g < a.group; ++g) {
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] |= *reinterpret_cast<const u32x4 *>(a.in + (bc[u] * a.group + g) * 128 + 4 * k);
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = reduce_load(reinterpret_cast<const u32x4 *>(a.in + bc[u] * 128 + 4 * k));
    }
    if constexpr (WHAT == RED_MAXBITSD1) {
      // explicit offsets, or chained from the input: block k starts from in[128k-1], block 0 from seed0.  Chained,
      // only the wave's first block reads it from memory: the others find it in the wave (below)
      if (a.seeds) {
#pragma unroll
        for (int u = 0; u < U; ++u) seed[u] = a.seeds[bc[u]];
      } else {
        const uint32_t s0 = a.in[bc[0] ? bc[0] * 128 - 1 : 0];
        seed[0] = bc[0] ? s0 : a.seed0;
      }
    }
    if constexpr (WHAT == RED_MAXBITSD1) {
      if (!a.seeds) {
        // the wave's blocks are consecutive, even ones in the low half: an odd block's predecessor ends in lane 31
        // of the same step, an even block's in lane 63 of the step before
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t lo_last = (uint32_t)__builtin_amdgcn_readlane((int)x[u].w, 31);
          const uint32_t up_last = u ? (uint32_t)__builtin_amdgcn_readlane((int)x[u - 1].w, 63) : seed[0];
          seed[u] = h ? lo_last : up_last;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (WHAT == RED_MINMAX) {
        // every row's minimum and maximum go to LDS, [half-wave][step][row]; the reduction over the rows follows
        // the loop, for all of the wave's blocks at once
        minmax_rows[0][wv][h][u][k] = min(min(x[u].x, x[u].y), min(x[u].z, x[u].w));
        minmax_rows[1][wv][h][u][k] = max(max(x[u].x, x[u].y), max(x[u].z, x[u].w));
      } else {
        u32x4 v = x[u];
        if constexpr (WHAT == RED_MAXBITSD1) {
          uint32_t prev = dpp0<DPP_WAVE_SHR1>(v.w);
          if (k == 0) prev = seed[u];
          v = u32x4{x[u].x - prev, x[u].y - x[u].x, x[u].z - x[u].y, x[u].w - x[u].z};
        }
        uint32_t t = v.x | v.y | v.z | v.w;
        // OR-reduce into lane 31 of the half-wave
        t |= dpp0<DPP_ROW_SHR1>(t);
        t |= dpp0<DPP_ROW_SHR2>(t);
        t |= dpp0<DPP_ROW_SHR4>(t);
        t |= dpp0<DPP_ROW_SHR8>(t);
        t |= dpp0<DPP_ROW_BCAST15, 0xA>(t);
        // results gathered into lanes 0 .. 2U-1 (block 2u + h of the wave in lane 2u + h): one store per array
        const uint32_t w0 = bits_of((uint32_t)__builtin_amdgcn_readlane((int)t, 31));
        const uint32_t w1 = bits_of((uint32_t)__builtin_amdgcn_readlane((int)t, 63));
        out_bits = lane == 2 * u ? w0 : (lane == 2 * u + 1 ? w1 : out_bits);
        if constexpr (WHAT == RED_MAXBITSD1) {
          const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)seed[u], 0);
          const uint32_t s1 = (uint32_t)__builtin_amdgcn_readlane((int)seed[u], 32);
          out_seed = lane == 2 * u ? s0 : (lane == 2 * u + 1 ? s1 : out_seed);
        }
      }
    }
    if constexpr (WHAT == RED_MINMAX) {
      // 64 / (2 U) lanes per block: each takes the minimum and maximum of 32 / that many consecutive rows (one
      // 16-byte LDS read per array), then the lanes of a block combine through DPP moves that stay inside their group
      // of eight (neighbour, other pair, other quad).  A reduction per block and array across 32 lanes -- five DPP
      // steps each, with a register copy per step because min and max have no neutral fill value -- cost this
      // kernel 8 points of HBM peak against the OR-reductions of the width kernels.
      static_assert(U == 4, "eight lanes per block, four rows per lane");
      wave_lds_fence();
      const int i = lane >> 3, part = lane & 7;  // block 2 * uu + hh of the wave
      const int hh = i / U, uu = i % U;
      const u32x4 l4 = *reinterpret_cast<const u32x4 *>(&minmax_rows[0][wv][hh][uu][4 * part]);
      const u32x4 h4 = *reinterpret_cast<const u32x4 *>(&minmax_rows[1][wv][hh][uu][4 * part]);
      uint32_t lo = min(min(l4.x, l4.y), min(l4.z, l4.w));
      uint32_t hi = max(max(h4.x, h4.y), max(h4.z, h4.w));
      constexpr int QUAD_SWAP1 = 0xB1, QUAD_SWAP2 = 0x4E, ROW_HALF_MIRROR = 0x141;
      lo = min(lo, dpp_keep<QUAD_SWAP1>(lo));
      hi = max(hi, dpp_keep<QUAD_SWAP1>(hi));
      lo = min(lo, dpp_keep<QUAD_SWAP2>(lo));
      hi = max(hi, dpp_keep<QUAD_SWAP2>(hi));
      lo = min(lo, dpp_keep<ROW_HALF_MIRROR>(lo));
      hi = max(hi, dpp_keep<ROW_HALF_MIRROR>(hi));
      const size_t mine = tile * per_iter + (size_t)wv * 2 * U + (size_t)(2 * uu + hh);
      if (part == 0 && mine < a.nblocks) {
        if (a.min_out) a.min_out[mine] = lo;
        if (a.max_out) a.max_out[mine] = hi;
        if (a.bits_out) a.bits_out[mine] = (uint8_t)bits_of(hi - lo);
      }
    } else {
      const size_t mine = tile * per_iter + (size_t)wv * 2 * U + (size_t)lane;
      if (lane < 2 * U && mine < a.nblocks) {
        {
          a.bits_out[mine] = (uint8_t)out_bits;
          if (WHAT == RED_MAXBITSD1 && a.seeds_out) a.seeds_out[mine] = out_seed;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// exclusive scan of 16 * bits[k] -> payload byte offsets (three small passes)
// ---------------------------------------------------------------------------
constexpr int SCAN_TILE = 4096;  // blocks per workgroup: 256 threads x 16

__device__ __forceinline__ uint64_t block_reduce_sum(uint64_t v, uint64_t *sh) {
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor((unsigned long long)v, d, 64);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[wv] = v;
  __syncthreads();
  uint64_t tot = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return tot;
}

// A tile is 4096 widths: wave w takes 1024 consecutive ones in 16 rounds of 64, lane l the width at 64 * round + l, so
// that every load and store is one coalesced instruction and all 16 loads go out back to back (widths past the end
// re-read the last one and count as zero).
__device__ __forceinline__ void scan_load_widths(const uint8_t *bits, size_t n, size_t wave_base, int lane, uint32_t (&v)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const size_t i = wave_base + 64 * (size_t)r + (size_t)lane;
    v[r] = bits[i < n ? i : n - 1];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = wave_base + 64 * (size_t)r + (size_t)lane < n ? v[r] : 0u;
}

__global__ __launch_bounds__(256) void k_scan_tile_sums(const uint8_t *bits, size_t n, uint64_t *tile_sums) {
  __shared__ uint64_t sh[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t v[16];
  scan_load_widths(bits, n, (size_t)blockIdx.x * SCAN_TILE + (size_t)wv * 1024, lane, v);
  uint64_t s = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += v[r];
  const uint64_t tot = block_reduce_sum(s, sh);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot * 16;
}

// one workgroup: exclusive scan of the tile sums in place; total -> *total_out
__global__ __launch_bounds__(256) void k_scan_spine(uint64_t *tile_sums, size_t ntiles, uint64_t *total_out) {
  __shared__ uint64_t sh[256];
  uint64_t carry = 0;
  for (size_t base = 0; base < ntiles; base += 256) {
    const size_t i = base + threadIdx.x;
    const uint64_t v = i < ntiles ? tile_sums[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const uint64_t t = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < ntiles) tile_sums[i] = carry + sh[threadIdx.x] - v;
    carry += sh[255];
    __syncthreads();
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

__global__ __launch_bounds__(256) void k_scan_apply(const uint8_t *bits, size_t n, const uint64_t *tile_sums,
                                                    uint64_t *off) {
  __shared__ uint64_t sh[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t wave_base = (size_t)blockIdx.x * SCAN_TILE + (size_t)wv * 1024;
  uint32_t v[16];
  scan_load_widths(bits, n, wave_base, lane, v);
  const uint64_t tile_in = tile_sums[blockIdx.x];
  // inclusive scans of the 16 rounds (sums of 64 widths fit 32 bits), then the rounds chained through lane 63
  uint32_t inc[16];
  uint64_t before[16];
  uint64_t run = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    uint32_t t = v[r];
    t += dpp0<DPP_ROW_SHR1>(t);
    t += dpp0<DPP_ROW_SHR2>(t);
    t += dpp0<DPP_ROW_SHR4>(t);
    t += dpp0<DPP_ROW_SHR8>(t);
    t += dpp0<DPP_ROW_BCAST15, 0xA>(t);
    t += dpp0<DPP_ROW_BCAST31, 0xC>(t);
    inc[r] = t;
    before[r] = run;
    run += (uint32_t)__builtin_amdgcn_readlane((int)t, 63);
  }
  if (lane == 0) sh[wv] = run;
  __syncthreads();
  uint64_t waves_before = 0;
  for (int w = 0; w < wv; ++w) waves_before += sh[w];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const size_t i = wave_base + 64 * (size_t)r + (size_t)lane;
    if (i < n) off[i] = tile_in + 16 * (waves_before + before[r] + inc[r] - v[r]);
  }
}

// ---------------------------------------------------------------------------
// verification helpers: position-sensitive 64-bit checksum and mismatch count
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// acc[0] += sum over words i of mix64((index0 + i) << 32 | word[i]); when `other`
// is given, acc[1] += number of words that differ from other[i].
__global__ __launch_bounds__(256) void k_checksum(const uint32_t *data, const uint32_t *other, size_t nwords,
                                                  uint64_t index0, unsigned long long *acc) {
  __shared__ uint64_t sh[4];
  uint64_t s = 0, bad = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) {
    const uint32_t v = data[i];
    s += mix64(((index0 + i) << 32) ^ v);
    if (other) bad += (other[i] != v);
  }
  const uint64_t tot = block_reduce_sum(s, sh);
  const uint64_t totbad = block_reduce_sum(bad, sh);
  if (threadIdx.x == 0) {
    atomicAdd(&acc[0], (unsigned long long)tot);
    if (other && totbad) atomicAdd(&acc[1], (unsigned long long)totbad);
  }
}

// ---------------------------------------------------------------------------
// whole-array reductions for the *_length routines: r[0] = OR of the values,
// r[1] = OR of successive differences seeded by `init`, r[2] = min, r[3] = max
// ---------------------------------------------------------------------------
__global__ void k_reduce_array_init(uint32_t *r) {
  r[0] = 0;
  r[1] = 0;
  r[2] = 0xFFFFFFFFu;
  r[3] = 0;
}

// What a workgroup's lanes hold goes into the four results: butterflies inside the waves, the waves combined through
// LDS, then at most one atomic per result and workgroup -- none where the result already covers the workgroup's
// share (results only ever gain bits / move outwards, so a stale look at them can cost an atomic, never lose one):
// half a million workgroups hammering four words with atomics would take longer than reading the array.
__device__ __forceinline__ void reduce_array_commit(uint32_t o, uint32_t od, uint32_t lo, uint32_t hi, uint32_t *r) {
  __shared__ uint32_t part[4][4];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    o |= (uint32_t)__shfl_xor((int)o, d, 64);
    od |= (uint32_t)__shfl_xor((int)od, d, 64);
    lo = min(lo, (uint32_t)__shfl_xor((int)lo, d, 64));
    hi = max(hi, (uint32_t)__shfl_xor((int)hi, d, 64));
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    part[wv][0] = o;
    part[wv][1] = od;
    part[wv][2] = lo;
    part[wv][3] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    o = part[0][0] | part[1][0] | part[2][0] | part[3][0];
    od = part[0][1] | part[1][1] | part[2][1] | part[3][1];
    lo = min(min(part[0][2], part[1][2]), min(part[2][2], part[3][2]));
    hi = max(max(part[0][3], part[1][3]), max(part[2][3], part[3][3]));
    const uint32_t c0 = __hip_atomic_load(&r[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t c1 = __hip_atomic_load(&r[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t c2 = __hip_atomic_load(&r[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t c3 = __hip_atomic_load(&r[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (o & ~c0) atomicOr(&r[0], o);
    if (od & ~c1) atomicOr(&r[1], od);
    if (lo < c2) atomicMin(&r[2], lo);
    if (hi > c3) atomicMax(&r[3], hi);
  }
}

// The bulk of a 16-byte aligned array: a workgroup takes REDUCE_ARRAY_U * 4 KiB and ends (one-shot workgroups, all
// loads of a wave issued together, non-temporal: see the block kernels); a lane's predecessor is the last value of
// the lane below (DPP), of the wave's previous step (lane 63), or, for the wave's first vector, one word from memory.
constexpr int REDUCE_ARRAY_U = 8;

__global__ __launch_bounds__(256) void k_reduce_array_bulk(const uint32_t *in, uint32_t init, uint32_t *r) {
  constexpr int U = REDUCE_ARRAY_U;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // wave w of workgroup g: vectors [(g * 4 + w) * U * 64, + U * 64), step u the 64 consecutive ones from u * 64 on
  const size_t v0 = ((size_t)blockIdx.x * 4 + wv) * U * 64;
  const u32x4 *in4 = reinterpret_cast<const u32x4 *>(in) + v0;
  u32x4 x[U];
#pragma unroll
  for (int u = 0; u < U; ++u) x[u] = reduce_load(in4 + u * 64 + lane);
  const uint32_t before = v0 ? in[4 * v0 - 1] : init;
  uint32_t o = 0, od = 0, lo = 0xFFFFFFFFu, hi = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const u32x4 v = x[u];
    uint32_t prev = dpp0<DPP_WAVE_SHR1>(v.w);
    const uint32_t first_prev = u ? (uint32_t)__builtin_amdgcn_readlane((int)x[u ? u - 1 : 0].w, 63) : before;
    prev = lane ? prev : first_prev;
    o |= v.x | v.y | v.z | v.w;
    od |= (v.x - prev) | (v.y - v.x) | (v.z - v.y) | (v.w - v.z);
    lo = min(lo, min(min(v.x, v.y), min(v.z, v.w)));
    hi = max(hi, max(max(v.x, v.y), max(v.z, v.w)));
  }
  reduce_array_commit(o, od, lo, hi, r);
}

// Everything else: short arrays, the end behind the last whole workgroup of the bulk kernel, arrays that are only
// 4-byte aligned.  `first` is the index of the first element this launch looks at.
__global__ __launch_bounds__(256) void k_reduce_array(const uint32_t *in, size_t first, size_t n, uint32_t init, uint32_t *r) {
  uint32_t o = 0, od = 0, lo = 0xFFFFFFFFu, hi = 0;
  const size_t stride = (size_t)gridDim.x * 256, gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool vec = (reinterpret_cast<uintptr_t>(in) & 15u) == 0 && (first & 3) == 0;
  const size_t nvec = vec ? n / 4 : first / 4;
  const u32x4 *in4 = reinterpret_cast<const u32x4 *>(in);
  for (size_t j = first / 4 + gid; j < nvec; j += stride) {
    const u32x4 v = in4[j];
    const uint32_t prev = j ? in[4 * j - 1] : init;
    o |= v.x | v.y | v.z | v.w;
    od |= (v.x - prev) | (v.y - v.x) | (v.z - v.y) | (v.w - v.z);
    lo = min(lo, min(min(v.x, v.y), min(v.z, v.w)));
    hi = max(hi, max(max(v.x, v.y), max(v.z, v.w)));
  }
  for (size_t i = (vec ? 4 * nvec : first) + gid; i < n; i += stride) {
    const uint32_t v = in[i], prev = i ? in[i - 1] : init;
    o |= v;
    od |= v - prev;
    lo = min(lo, v);
    hi = max(hi, v);
  }
  reduce_array_commit(o, od, lo, hi, r);
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
// one tile per workgroup (the kernels' tile loops then run once): see the note
// on one-shot workgroups above
static int grid_for_tiles(size_t tiles) {
  const size_t cap = (size_t)1 << 30;
  if (tiles > cap) tiles = cap;
  if (tiles < 1) tiles = 1;
  return (int)tiles;
}

static int grid_for(size_t nblocks, int U) {
  const size_t per_iter = (size_t)BLOCKS_PER_STEP * U;
  return grid_for_tiles((nblocks + per_iter - 1) / per_iter);
}

constexpr int FLEX_U = 2;
#ifndef SC_REDUCE_U
#define SC_REDUCE_U 4
#endif
constexpr int REDUCE_U = SC_REDUCE_U;

// Workgroup shape of the one-shot decoders, from a sweep of (threads, steps)
// at every width with 2^30 integers: the best shapes write 4 KiB (narrow
// widths, which are almost pure store traffic and need two blocks per half-wave
// to cover the payload read in front of the stores) or 2 KiB per workgroup;
// blocks of 20 bits and more prefer 256 threads.  The per-block-width kernels chase two
// dependent loads (offset, then payload) and measured best with four steps.
#ifndef SC_UN_BS
#define SC_UN_BS 128
#define SC_UN_U 2
#endif
#ifndef SC_UN_MAXB
#define SC_UN_MAXB 0
#endif
#ifndef SC_UW_MINB
#define SC_UW_MINB 99
#define SC_UW_BS 256
#define SC_UW_U 1
#endif
#ifndef SC_UM_LO
#define SC_UM_LO 99
#define SC_UM_HI 0
#define SC_UM_BS 128
#define SC_UM_U 1
#endif
#ifndef SC_MIXED_UNPACK_STEPS
#define SC_MIXED_UNPACK_STEPS 4
#endif
#ifndef SC_MIXED_UNPACK_BS
#define SC_MIXED_UNPACK_BS 256
#endif
constexpr int unpack_threads(int b) { return (b >= SC_UM_LO && b <= SC_UM_HI) ? SC_UM_BS : b >= SC_UW_MINB ? SC_UW_BS : (b < 0 ? SC_MIXED_UNPACK_BS : b >= 20 ? 256 : ((b >= 1 && b <= SC_UN_MAXB) ? SC_UN_BS : 128)); }
constexpr int unpack_steps(int b) { return (b >= SC_UM_LO && b <= SC_UM_HI) ? SC_UM_U : b >= SC_UW_MINB ? SC_UW_U : (b < 0 ? SC_MIXED_UNPACK_STEPS : ((b >= 1 && b <= SC_UN_MAXB) ? SC_UN_U : (b <= 14 ? 2 : 1))); }
// Widths 1..7 gain 3-17 points of HBM peak from dedicated prefetch workgroups
// (one per 16 decode workgroups, 20 MiB of output ahead: the plateau of a
// sweep over group size 8..512 and distance 4..64 MiB); from 8 bits on the
// payload is a large enough share of the traffic that the extra reads cost
// more than the latency they hide.  b = 1 is sharper than the others: on most boxes its rate
// peaks at 10 MiB (72 / 79 / 81 / 82 / 81 / 79 / 77 / 76 % at 4 / 6 / 8 / 10 / 12 / 14 / 16 / 20 MiB), on the
// others it runs at 85 % at either distance.  The per-block-width decoder gains 4-7 % from one prefetcher per 8
// decode workgroups that touches the metadata (widths, offsets, seeds, 60 MiB of output ahead): the decode waves'
// first round of loads then hits the cache and they wait for HBM once, not twice.  Half that distance ahead the
// metadata is in the cache for the prefetcher too, which reads where a group's payload begins and ends and touches
// it when the group is narrow (SC_PF_MIXED_PAYLOAD_MAXB above).
#ifndef SC_PF_GROUP
#define SC_PF_GROUP 16
#endif
#ifndef SC_PF_DIST_MIB
#define SC_PF_DIST_MIB 20
#endif
#ifndef SC_PF_MAXB
#define SC_PF_MAXB 7
#endif
#ifndef SC_PF_MIXED
#define SC_PF_MIXED 8
#endif
// (b = 5..7: one prefetcher per 24 decode workgroups, +1 point; 40 and more lose 3-5 points, as do 10 and 40 MiB)
#ifndef SC_PF_HI_MINB
#define SC_PF_HI_MINB 5
#define SC_PF_GROUP_HI 24
#define SC_PF_DIST_HI_MIB 20
#endif
constexpr int unpack_prefetch_group(int b) { return b < 0 ? SC_PF_MIXED : ((b >= 1 && b <= SC_PF_MAXB) ? (b >= SC_PF_HI_MINB ? SC_PF_GROUP_HI : SC_PF_GROUP) : 0); }
#ifndef SC_PF_DIST1_MIB
#define SC_PF_DIST1_MIB 10
#endif
#ifndef SC_PF_DISTM_MIB
#define SC_PF_DISTM_MIB 60
#endif
constexpr size_t prefetch_distance_bytes(int b) { return (size_t)(b < 0 ? SC_PF_DISTM_MIB : (b == 1 ? SC_PF_DIST1_MIB : (b >= SC_PF_HI_MINB ? SC_PF_DIST_HI_MIB : SC_PF_DIST_MIB))) << 20; }

// The encoders read 512 bytes per block whatever the width.  Shapes from sweeps at 2^30 integers with non-temporal
// input loads and payload stores (64 / 128 / 256 threads x 1 / 2 / 4 steps): 128 x 2 for b <= 12 and b = 16 (b = 8:
// 74 -> 83 % against 256 x 1), 256 x 1 for the other widths, 256 x 2 for per-block widths.
#ifndef SC_PM_BS
#define SC_PM_BS 256
#define SC_PM_U 2
#endif
#ifndef SC_PN_BS
#define SC_PN_BS 128
#define SC_PN_U 2
#endif
#ifndef SC_PW_LO
#define SC_PW_LO 99
#define SC_PW_HI 0
#define SC_PW_BS 256
#define SC_PW_U 1
#endif
constexpr bool pack_narrow_shape(int b) { return (b > 0 && b <= 12) || b == 16; }
constexpr int pack_threads(int b) { return (b >= SC_PW_LO && b <= SC_PW_HI) ? SC_PW_BS : (b < 0 ? SC_PM_BS : (pack_narrow_shape(b) ? (b >= 4 ? SC_PN_BS : 128) : 256)); }
constexpr int pack_steps(int b) { return (b >= SC_PW_LO && b <= SC_PW_HI) ? SC_PW_U : (b < 0 ? SC_PM_U : (pack_narrow_shape(b) ? (b >= 4 ? SC_PN_U : 2) : 1)); }

// (16-lane layout: 128 threads at every width; 256 from 20 bits on, as the 4-lane layout has them, ran at 72.5 %
// against 75-76 % at 24 and 32 bits.  A decoder with k_pack16's lane mapping -- whole wide blocks per wave, the payload
// fetched by one or two fully coalesced loads, groups taken apart in LDS -- was built and measured: 76-77 % at 24..32
// bits, 70-74 % below 16 bits where this kernel has its prefetch workgroups and runs at 78-83 %: not kept.)
#ifndef SC_U16_BS
#define SC_U16_BS 128
#endif
// index 0 is the per-block-width instance (B = -1), index 1 + b the uniform width b
template <int MODE, int BP1, int LS = 0>
static size_t launch_unpack_fast(const UnpackArgs &a, hipStream_t st) {
  // 16-lane layout: two steps, so that a wave's four blocks are the four groups of one wide block
  constexpr int U = LS == 2 ? 2 : unpack_steps(BP1 - 1), BS = (LS == 2 && SC_U16_BS) ? SC_U16_BS : unpack_threads(BP1 - 1), PF = unpack_prefetch_group(BP1 - 1);
  constexpr size_t per_wg = (size_t)(BS / 32) * U;
  const size_t nwg = a.nblocks / per_wg;
  if (nwg == 0) return 0;
  if constexpr (PF > 0) {
    const size_t groups = (nwg + PF - 1) / PF;
    const size_t dist = prefetch_distance_bytes(BP1 - 1) / (PF * per_wg * 512);
    hipLaunchKernelGGL((k_unpack_fast<BP1 - 1, MODE, U, BS, PF, LS>), dim3((unsigned)(groups * (PF + 1))), dim3(BS), 0, st, a,
                       (uint32_t)nwg, (uint32_t)dist);
  } else {
    hipLaunchKernelGGL((k_unpack_fast<BP1 - 1, MODE, U, BS, 0, LS>), dim3((unsigned)nwg), dim3(BS), 0, st, a, 0u, 0u);
  }
  return nwg * per_wg;
}

template <int MODE, int... Is>
static size_t dispatch_unpack_fast(int idx, const UnpackArgs &a, hipStream_t st, std::integer_sequence<int, Is...>) {
  using fn_t = size_t (*)(const UnpackArgs &, hipStream_t);
  static const fn_t table[] = {&launch_unpack_fast<MODE, Is>...};
  return table[idx](a, st);
}

// wide layouts: widths 0..32 (index = width), plain coding
template <int LS, int... Is>
static size_t dispatch_unpack_wide(int bit, const UnpackArgs &a, hipStream_t st, std::integer_sequence<int, Is...>) {
  using fn_t = size_t (*)(const UnpackArgs &, hipStream_t);
  static const fn_t table[] = {&launch_unpack_fast<MODE_PLAIN, Is + 1, LS>...};
  return table[bit](a, st);
}

static void launch_unpack_flex(int mode, bool tail, const UnpackArgs &a, hipStream_t st) {
  const dim3 g(grid_for(a.nblocks, FLEX_U)), t(256);
#define SC_RT(MODE_)                                                                         \
  if (tail) hipLaunchKernelGGL((k_unpack_flex<-1, MODE_, FLEX_U, true>), g, t, 0, st, a);     \
  else hipLaunchKernelGGL((k_unpack_flex<-1, MODE_, FLEX_U, false>), g, t, 0, st, a);
  if (mode == MODE_PLAIN) { SC_RT(MODE_PLAIN) }
  else if (mode == MODE_D1) { SC_RT(MODE_D1) }
  else { SC_RT(MODE_FOR) }
#undef SC_RT
}

// one launch covers at most this many blocks (keeps grid.x far below 2^31 workgroups)
constexpr size_t MAX_BLOCKS_PER_LAUNCH = (size_t)1 << 30;

hipError_t launch_unpack(int mode, bool tail, const UnpackArgs &a, hipStream_t st) {
  if (a.nblocks == 0) return hipSuccess;
  const bool mixed = a.bits != nullptr && a.off != nullptr;
  const bool uniform = a.bits == nullptr && a.off == nullptr;
  const bool seeds_ok = mode == MODE_PLAIN || a.seeds != nullptr || uniform;  // uniform widths: seed0 serves every block
  const bool wide = a.lane_groups > 1;
  if (tail || !(mixed || uniform) || !seeds_ok || (wide && !(uniform && mode == MODE_PLAIN && (a.lane_groups == 2 || a.lane_groups == 4)))) {
    launch_unpack_flex(mode, tail, a, st);
    return hipGetLastError();
  }
  using seq = std::make_integer_sequence<int, 34>;
  using wseq = std::make_integer_sequence<int, 33>;
  const int idx = mixed ? 0 : 1 + (int)a.bit;
  UnpackArgs r = a;
  while (r.nblocks) {
    UnpackArgs c = r;
    if (c.nblocks > MAX_BLOCKS_PER_LAUNCH) c.nblocks = MAX_BLOCKS_PER_LAUNCH;
    size_t done;
    // (a workgroup's blocks are a multiple of four, so whole wide blocks: what is left over starts on one as well)
    if (wide) done = a.lane_groups == 2 ? dispatch_unpack_wide<1>((int)a.bit, c, st, wseq{}) : dispatch_unpack_wide<2>((int)a.bit, c, st, wseq{});
    else if (mode == MODE_PLAIN) done = dispatch_unpack_fast<MODE_PLAIN>(idx, c, st, seq{});
    else if (mode == MODE_D1) done = dispatch_unpack_fast<MODE_D1>(idx, c, st, seq{});
    else done = dispatch_unpack_fast<MODE_FOR>(idx, c, st, seq{});
    if (done == 0) {  // fewer blocks than one workgroup handles: ragged end
      launch_unpack_flex(mode, false, r, st);
      break;
    }
    r.nblocks -= done;
    r.length = r.nblocks * 128;
    r.out += done * 128;
    if (r.seeds) r.seeds += done;
    if (mixed) {
      r.bits += done;
      r.off += done;
      r.block_base += done;
    } else {
      r.payload = static_cast<const uint8_t *>(r.payload) + done * 16 * (size_t)a.bit;
    }
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// The 16-lane layout's own encoder (avx512pack / avx512packwithoutmask, src/avx512bitpacking.c): a wave takes whole
// wide blocks of 512 integers.  Lane 16 g + r holds rows r and r + 16 of lane group g (integers 16 R + 4 g .. + 3 of
// the block for row R), so that one load instruction of the wave reads 16 rows x 64 bytes = 1 KiB in one piece (with
// one four-lane group per half-wave, as k_pack_fast maps it, an instruction touches 32 bytes of every 64-byte line
// and each line is fetched by two instructions: 71-78 % of the HBM peak where the 8-lane layout, whose two groups
// do fill a line, reaches 80-84 %).  A lane group's rows 0..15 and 16..31 each sit in one DPP row of 16 lanes: the
// hand-over of spilled bits and the segmented OR-scan over the rows that share an output word use row shifts inside
// the two halves and a row rotation from the first half into the second (row 15 -> row 16 and on).  The four groups'
// vectors of one output word are 64 contiguous bytes, stored by one instruction.
// ---------------------------------------------------------------------------
constexpr int DPP_ROW_ROR1 = 0x121, DPP_ROW_ROR2 = 0x122, DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128;
#ifndef SC_P16_BS
#define SC_P16_BS 128
#endif
#ifndef SC_P16_U
#define SC_P16_U 1
#endif
#ifndef SC_P16
#define SC_P16 1
#endif

template <int B, bool MASKED, int U, int BS>
__global__ __launch_bounds__(BS) void k_pack16(PackArgs a) {
  static_assert(B >= 1 && B <= 32, "widths 1..32");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t g = (uint32_t)lane >> 4, r = (uint32_t)lane & 15;
  const u32x4 *in = reinterpret_cast<const u32x4 *>(a.in);
  u32x4 *payload = reinterpret_cast<u32x4 *>(a.payload);
  const size_t first = ((size_t)blockIdx.x * (BS / 64) + (size_t)wv) * U;
  u32x4 x0[U], x1[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const u32x4 *blk = in + (first + u) * 128 + 4 * r + g;
    x0[u] = __builtin_nontemporal_load(blk);
    x1[u] = __builtin_nontemporal_load(blk + 64);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    u32x4 *out = payload + (first + u) * (size_t)(4 * B) + g;  // vector w of this group at out + 4 w
    if constexpr (B == 32) {
      __builtin_nontemporal_store(x0[u], out + 4 * r);
      __builtin_nontemporal_store(x1[u], out + 4 * (r + 16));
    } else {
      u32x4 v0 = x0[u], v1 = x1[u];
      if constexpr (MASKED) {
        v0 &= field_mask(B);
        v1 &= field_mask(B);
      }
      const uint32_t p0 = r * B, w0 = p0 >> 5, s0 = p0 & 31;
      const uint32_t p1 = (r + 16) * B, w1 = p1 >> 5, s1 = p1 & 31;
      u32x4 t0 = v0 << s0, t1 = v1 << s1;
      // bits that do not fit the row's first word go to the next row's word (which is where that row starts)
      const u32x4 sp0 = (v0 >> ((32 - s0) & 31)) & (s0 + B > 32 ? 0xFFFFFFFFu : 0u);
      const u32x4 sp1 = (v1 >> ((32 - s1) & 31)) & (s1 + B > 32 ? 0xFFFFFFFFu : 0u);
      const u32x4 sp0_last = sp0 & (r == 15 ? 0xFFFFFFFFu : 0u);  // row 15 hands over to row 16
      t0.x |= dpp0<DPP_ROW_SHR1>(sp0.x);
      t0.y |= dpp0<DPP_ROW_SHR1>(sp0.y);
      t0.z |= dpp0<DPP_ROW_SHR1>(sp0.z);
      t0.w |= dpp0<DPP_ROW_SHR1>(sp0.w);
      t1.x |= dpp0<DPP_ROW_SHR1>(sp1.x) | dpp0<DPP_ROW_ROR1>(sp0_last.x);
      t1.y |= dpp0<DPP_ROW_SHR1>(sp1.y) | dpp0<DPP_ROW_ROR1>(sp0_last.y);
      t1.z |= dpp0<DPP_ROW_SHR1>(sp1.z) | dpp0<DPP_ROW_ROR1>(sp0_last.z);
      t1.w |= dpp0<DPP_ROW_SHR1>(sp1.w) | dpp0<DPP_ROW_ROR1>(sp0_last.w);
      // Segmented inclusive OR-scan over the rows that start in the same word.  Row R hands its partial word to row
      // R + D when that row starts in the same word, i.e. D * B <= bit offset of row R + D (masked at the giving
      // lane, so that the taker's OR folds the DPP move in).  Rows 16 - D .. 15 give to the second half (rotation).
#define SC_P16_STEP(D, SHR, ROR)                                                                            \
  if constexpr (D < max_rows_per_word(B)) {                                                                 \
    const uint32_t give0 = ((uint32_t)D * B <= (((r + D) * B) & 31u)) ? 0xFFFFFFFFu : 0u;                    \
    const uint32_t give1 = ((uint32_t)D * B <= (((r + 16 + D) * B) & 31u)) ? 0xFFFFFFFFu : 0u;               \
    const uint32_t wrap = (r >= 16 - D) ? give0 : 0u;                                                       \
    const u32x4 n1 = {t1.x | dpp0<SHR>(t1.x & give1) | dpp0<ROR>(t0.x & wrap), t1.y | dpp0<SHR>(t1.y & give1) | dpp0<ROR>(t0.y & wrap), \
                      t1.z | dpp0<SHR>(t1.z & give1) | dpp0<ROR>(t0.z & wrap), t1.w | dpp0<SHR>(t1.w & give1) | dpp0<ROR>(t0.w & wrap)}; \
    t0.x |= dpp0<SHR>(t0.x & give0);                                                                         \
    t0.y |= dpp0<SHR>(t0.y & give0);                                                                         \
    t0.z |= dpp0<SHR>(t0.z & give0);                                                                         \
    t0.w |= dpp0<SHR>(t0.w & give0);                                                                         \
    t1 = n1;                                                                                                \
  }
      SC_P16_STEP(1, DPP_ROW_SHR1, DPP_ROW_ROR1)
      SC_P16_STEP(2, DPP_ROW_SHR2, DPP_ROW_ROR2)
      SC_P16_STEP(4, DPP_ROW_SHR4, DPP_ROW_ROR4)
      SC_P16_STEP(8, DPP_ROW_SHR8, DPP_ROW_ROR8)
#undef SC_P16_STEP
      if constexpr (16 < max_rows_per_word(B)) t1 |= t0;  // one bit per value: all 32 rows share word 0
      // the last row of a word holds it whole
      const bool last0 = (((r + 1) * B) >> 5) != w0;
      const bool last1 = r == 15 || (((r + 17) * B) >> 5) != w1;
      if (last0) __builtin_nontemporal_store(t0, out + 4 * w0);
      if (last1) __builtin_nontemporal_store(t1, out + 4 * w1);
    }
  }
}

// -> blocks done, in four-lane blocks (four per wide block)
template <bool MASKED, int BP1>
static size_t launch_pack16(const PackArgs &a, hipStream_t st) {
  constexpr int B = BP1 - 1;
  if constexpr (B < 1) {
    return 0;
  } else {
    constexpr size_t per_wg = (size_t)(SC_P16_BS / 64) * SC_P16_U;  // wide blocks per workgroup
    const size_t nwg = (a.nblocks / 4) / per_wg;
    if (nwg) hipLaunchKernelGGL((k_pack16<B, MASKED, SC_P16_U, SC_P16_BS>), dim3((unsigned)nwg), dim3(SC_P16_BS), 0, st, a);
    return nwg * per_wg * 4;
  }
}

template <bool MASKED, int... Is>
static size_t dispatch_pack16(int bit, const PackArgs &a, hipStream_t st, std::integer_sequence<int, Is...>) {
  using fn_t = size_t (*)(const PackArgs &, hipStream_t);
  static const fn_t table[] = {&launch_pack16<MASKED, Is + 1>...};
  return table[bit](a, st);
}

template <int MODE, bool MASKED, int BP1, int LS = 0, bool GENERIC = false>
static size_t launch_pack_fast(const PackArgs &a, hipStream_t st) {
  constexpr int U = pack_steps(BP1 - 1), BS = pack_threads(BP1 - 1);
  constexpr size_t per_wg = (size_t)(BS / 32) * U;
  const size_t nwg = a.nblocks / per_wg;
  if (nwg) hipLaunchKernelGGL((k_pack_fast<BP1 - 1, MODE, MASKED, U, BS, LS, GENERIC>), dim3((unsigned)nwg), dim3(BS), 0, st, a);
  return nwg * per_wg;
}

template <int MODE, bool MASKED, int... Is>
static size_t dispatch_pack_fast(int idx, const PackArgs &a, hipStream_t st, std::integer_sequence<int, Is...>) {
  using fn_t = size_t (*)(const PackArgs &, hipStream_t);
  static const fn_t table[] = {&launch_pack_fast<MODE, MASKED, Is>...};
  return table[idx](a, st);
}

template <bool MASKED, int LS, int... Is>
static size_t dispatch_pack_wide(int bit, const PackArgs &a, hipStream_t st, std::integer_sequence<int, Is...>) {
  using fn_t = size_t (*)(const PackArgs &, hipStream_t);
  static const fn_t table[] = {&launch_pack_fast<MODE_PLAIN, MASKED, Is + 1, LS>...};
  return table[bit](a, st);
}

// full blocks of simdpackFOR_length: widths 1..31 (index = width - 1)
template <int... Is>
static size_t dispatch_pack_generic_for(int bit, const PackArgs &a, hipStream_t st, std::integer_sequence<int, Is...>) {
  using fn_t = size_t (*)(const PackArgs &, hipStream_t);
  static const fn_t table[] = {&launch_pack_fast<MODE_FOR, false, Is + 2, 0, true>...};
  return table[bit - 1](a, st);
}

static void launch_pack_flex(int mode, bool masked, bool generic, const PackArgs &a, hipStream_t st) {
  const dim3 g(grid_for(a.nblocks, FLEX_U)), t(256);
  if (generic) {
    // arbitrary-length packer: unmasked by definition (plain or frame-of-reference)
    if (mode == MODE_FOR) hipLaunchKernelGGL((k_pack_flex<-1, MODE_FOR, false, FLEX_U, true>), g, t, 0, st, a);
    else hipLaunchKernelGGL((k_pack_flex<-1, MODE_PLAIN, false, FLEX_U, true>), g, t, 0, st, a);
    return;
  }
#define SC_RT(MODE_)                                                                           \
  if (masked) hipLaunchKernelGGL((k_pack_flex<-1, MODE_, true, FLEX_U, false>), g, t, 0, st, a); \
  else hipLaunchKernelGGL((k_pack_flex<-1, MODE_, false, FLEX_U, false>), g, t, 0, st, a);
  if (mode == MODE_PLAIN) { SC_RT(MODE_PLAIN) }
  else if (mode == MODE_D1) { SC_RT(MODE_D1) }
  else { hipLaunchKernelGGL((k_pack_flex<-1, MODE_FOR, false, FLEX_U, false>), g, t, 0, st, a); }
#undef SC_RT
}

hipError_t launch_pack(int mode, bool masked, bool generic, const PackArgs &a, hipStream_t st) {
  if (a.nblocks == 0) return hipSuccess;
  const bool mixed = a.bits != nullptr && a.off != nullptr;
  const bool uniform = a.bits == nullptr && a.off == nullptr;
  const bool seeds_ok = mode != MODE_FOR || a.seeds != nullptr || uniform;  // uniform widths: seed0 serves every block
  if (mode == MODE_FOR) masked = false;
  const bool wide = a.lane_groups > 1;
  if (generic && mode == MODE_FOR && uniform && !a.seeds && !wide && a.carry_blocks && a.bit >= 1 && a.bit <= 31 && a.length >= 128) {
    // simdpackFOR_length: the full blocks through the fast kernels, what is left (ragged end of the workgroup
    // tiling, partial last block) through the flexible kernel, which is told that blocks lie in front of it
    PackArgs r = a;
    size_t full = a.length / 128;
    while (full) {
      PackArgs c = r;
      c.nblocks = full > MAX_BLOCKS_PER_LAUNCH ? MAX_BLOCKS_PER_LAUNCH : full;
      c.length = c.nblocks * 128;
      const size_t done = dispatch_pack_generic_for((int)a.bit, c, st, std::make_integer_sequence<int, 31>{});
      if (done == 0) break;
      full -= done;
      r.nblocks -= done;
      r.length -= done * 128;
      r.in += done * 128;
      r.carry_first = true;
      if (r.seeds) r.seeds += done;
      r.payload = static_cast<uint8_t *>(r.payload) + done * 16 * (size_t)a.bit;
    }
    if (r.nblocks) launch_pack_flex(mode, masked, true, r, st);
    return hipGetLastError();
  }
  if (generic || !(mixed || uniform) || !seeds_ok || (wide && !(uniform && mode == MODE_PLAIN && (a.lane_groups == 2 || a.lane_groups == 4)))) {
    launch_pack_flex(mode, masked, generic, a, st);
    return hipGetLastError();
  }
  using seq = std::make_integer_sequence<int, 34>;
  using wseq = std::make_integer_sequence<int, 33>;
  const int idx = mixed ? 0 : 1 + (int)a.bit;
  PackArgs r = a;
  while (r.nblocks) {
    PackArgs c = r;
    if (c.nblocks > MAX_BLOCKS_PER_LAUNCH) c.nblocks = MAX_BLOCKS_PER_LAUNCH;
    size_t done;
    if (wide) {
      if (a.lane_groups == 2) done = masked ? dispatch_pack_wide<true, 1>((int)a.bit, c, st, wseq{}) : dispatch_pack_wide<false, 1>((int)a.bit, c, st, wseq{});
      else if (SC_P16 && a.bit >= 1) done = masked ? dispatch_pack16<true>((int)a.bit, c, st, wseq{}) : dispatch_pack16<false>((int)a.bit, c, st, wseq{});
      else done = masked ? dispatch_pack_wide<true, 2>((int)a.bit, c, st, wseq{}) : dispatch_pack_wide<false, 2>((int)a.bit, c, st, wseq{});
    } else if (mode == MODE_PLAIN) {
      done = masked ? dispatch_pack_fast<MODE_PLAIN, true>(idx, c, st, seq{})
                    : dispatch_pack_fast<MODE_PLAIN, false>(idx, c, st, seq{});
    } else if (mode == MODE_D1) {
      done = masked ? dispatch_pack_fast<MODE_D1, true>(idx, c, st, seq{})
                    : dispatch_pack_fast<MODE_D1, false>(idx, c, st, seq{});
    } else {
      done = dispatch_pack_fast<MODE_FOR, false>(idx, c, st, seq{});
    }
    if (done == 0) {
      launch_pack_flex(mode, masked, false, r, st);
      break;
    }
    r.nblocks -= done;
    r.length = r.nblocks * 128;
    r.in += done * 128;
    r.chain_base += done;
    if (r.seeds) r.seeds += done;
    if (r.seeds_out) r.seeds_out += done;
    if (mixed) {
      r.bits += done;
      r.off += done;
    } else {
      r.payload = static_cast<uint8_t *>(r.payload) + done * 16 * (size_t)a.bit;
    }
  }
  return hipGetLastError();
}

hipError_t launch_reduce(int what, const ReduceArgs &a, hipStream_t st) {
  if (a.nblocks == 0) return hipSuccess;
  const dim3 g(grid_for(a.nblocks, REDUCE_U)), t(256);
  if (what == RED_MAXBITS) hipLaunchKernelGGL((k_reduce<RED_MAXBITS, REDUCE_U>), g, t, 0, st, a);
  else if (what == RED_MAXBITSD1) hipLaunchKernelGGL((k_reduce<RED_MAXBITSD1, REDUCE_U>), g, t, 0, st, a);
  else hipLaunchKernelGGL((k_reduce<RED_MINMAX, REDUCE_U>), g, t, 0, st, a);
  return hipGetLastError();
}

size_t scan_workspace_bytes(size_t nblocks) {
  const size_t ntiles = (nblocks + SCAN_TILE - 1) / SCAN_TILE;
  return (ntiles ? ntiles : 1) * sizeof(uint64_t);
}

hipError_t launch_payload_offsets(const uint8_t *bits, size_t nblocks, uint64_t *off, uint64_t *total,
                                  void *workspace, hipStream_t st) {
  uint64_t *tile_sums = reinterpret_cast<uint64_t *>(workspace);
  const size_t ntiles = (nblocks + SCAN_TILE - 1) / SCAN_TILE;
  if (nblocks == 0) {
    if (total) return hipMemsetAsync(total, 0, sizeof(uint64_t), st);
    return hipSuccess;
  }
  hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)ntiles), dim3(256), 0, st, bits, nblocks, tile_sums);
  hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(256), 0, st, tile_sums, ntiles, total);
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)ntiles), dim3(256), 0, st, bits, nblocks, tile_sums, off);
  return hipGetLastError();
}

hipError_t launch_checksum(const uint32_t *data, const uint32_t *other, size_t nwords, uint64_t index0,
                           uint64_t *acc, hipStream_t st) {
  if (nwords == 0) return hipSuccess;
  size_t g = (nwords + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_checksum, dim3((unsigned)g), dim3(256), 0, st, data, other, nwords, index0,
                     reinterpret_cast<unsigned long long *>(acc));
  return hipGetLastError();
}

hipError_t launch_reduce_array(const uint32_t *in, size_t length, uint32_t init, uint32_t *result4, hipStream_t st) {
  hipLaunchKernelGGL(k_reduce_array_init, dim3(1), dim3(1), 0, st, result4);
  if (length) {
    // whole workgroups of the bulk kernel first (aligned arrays), the generic kernel for what is left
    constexpr size_t per_wg = (size_t)REDUCE_ARRAY_U * 256 * 4;  // integers
    size_t first = 0;
    if ((reinterpret_cast<uintptr_t>(in) & 15u) == 0 && length >= per_wg) {
      size_t nwg = length / per_wg;
      if (nwg > ((size_t)1 << 30)) nwg = (size_t)1 << 30;  // (grid limit; the generic kernel takes the rest)
      hipLaunchKernelGGL(k_reduce_array_bulk, dim3((unsigned)nwg), dim3(256), 0, st, in, init, result4);
      first = nwg * per_wg;
    }
    if (first < length) {
      size_t g = ((length - first) / 4 + 255) / 256;
      if (g > 2048) g = 2048;
      if (g < 1) g = 1;
      hipLaunchKernelGGL(k_reduce_array, dim3((unsigned)g), dim3(256), 0, st, in, first, length, init, result4);
    }
  }
  return hipGetLastError();
}

}  // namespace sc
