// Lane-level building blocks shared by the kernel files: the 16-byte vector type, DPP moves inside a wave64,
// funnel shifts, half-wave scans, streaming accesses and the wave-private LDS hand-off.  Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sc {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// The same 16 bytes at any byte address: global memory runs in unaligned mode on this platform, so this is still one
// global_load_dwordx4 / global_store_dwordx4 (tools/exp/unaligned_exp.hip checks it).  LDS accesses stay aligned.
typedef u32x4 u32x4_u __attribute__((aligned(1)));
// A wave works on 2 * U consecutive blocks: block 2u + h belongs to half-wave h at step u.  Per-block metadata is
// fetched with ONE coalesced load per array (lane l holds the entry of block l mod 2U) and handed to its half-wave
// through scalar registers: 2U / 1 fewer memory instructions than every lane loading its own block's entry.
template <int U>
__device__ __forceinline__ void hand_out(uint32_t mine, int h, uint32_t (&v)[U]) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t v0 = (uint32_t)__builtin_amdgcn_readlane((int)mine, 2 * u);
    const uint32_t v1 = (uint32_t)__builtin_amdgcn_readlane((int)mine, 2 * u + 1);
    v[u] = h ? v1 : v0;
  }
}
template <int U>
__device__ __forceinline__ void hand_out(uint64_t mine, int h, uint64_t (&v)[U]) {
  uint32_t lo[U], hi[U];
  hand_out<U>((uint32_t)mine, h, lo);
  hand_out<U>((uint32_t)(mine >> 32), h, hi);
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = ((uint64_t)hi[u] << 32) | lo[u];
}

// 16 readable bytes for loads that have nothing to read: an optional array that is absent, a zero-width block.  The
// load is issued all the same, so that it sits next to its neighbours instead of behind a branch of its own (after a
// branch the compiler waits for everything in flight before the next load goes out).
static __device__ u32x4 g_nothing;

// A load from an address that went through integer arithmetic: the cast back names the global address space, so the
// compiler emits a global load (a generic pointer gives a flat load, which also counts against the LDS wait counter).
template <typename T>
__device__ __forceinline__ T load_global_as(uintptr_t at) {
  return *(const __attribute__((address_space(1))) T *)at;
}
__device__ __forceinline__ u32x4 load_global_at(uintptr_t at) {
  return *(const __attribute__((address_space(1))) u32x4_u *)at;
}
// The same, non-temporal: for bytes that are read once and should not displace what the caches hold.
__device__ __forceinline__ u32x4 load_global_nt(uintptr_t at) {
  return __builtin_nontemporal_load((const __attribute__((address_space(1))) u32x4_u *)at);
}

constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142;
constexpr int DPP_ROW_BCAST31 = 0x143;
constexpr int DPP_WAVE_SHR1 = 0x138;

// Lane l receives `v` from the lane the DPP control names, or 0 where that lane
// does not exist / the row is masked off.
template <int CTRL, int ROWMASK = 0xF>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xF, false);
}

// The same, but a lane with no source (or in a masked-off row) keeps its own `v`: for reductions under min / max,
// which have no neutral element that DPP could fill in.
template <int CTRL, int ROWMASK = 0xF>
__device__ __forceinline__ uint32_t dpp_keep(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWMASK, 0xF, false);
}

__device__ __forceinline__ uint32_t field_mask(uint32_t b) { return (uint32_t)((1ull << b) - 1ull); }

__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t s) {
  return __builtin_amdgcn_alignbit(hi, lo, s);  // ({hi,lo} >> (s & 31))[31:0]
}

// Inclusive sum over the 32 lanes of each half-wave.
__device__ __forceinline__ uint32_t half_scan_add(uint32_t t) {
  t += dpp0<DPP_ROW_SHR1>(t);
  t += dpp0<DPP_ROW_SHR2>(t);
  t += dpp0<DPP_ROW_SHR4>(t);
  t += dpp0<DPP_ROW_SHR8>(t);
  t += dpp0<DPP_ROW_BCAST15, 0xA>(t);
  return t;
}

// Streaming accesses.  The decoded integers (and, for the encoders, the input
// integers) are touched once: a non-temporal store / load keeps them from
// displacing useful lines in L2 and the Infinity Cache.  Measured on the
// one-shot decoder: +5..+10 % at every width (SC_STREAM_STORES / SC_STREAM_LOADS
// = 0 restore plain accesses for A/B runs).
#ifndef SC_STREAM_STORES
#define SC_STREAM_STORES 1
#endif
#ifndef SC_STREAM_LOADS
#define SC_STREAM_LOADS 0
#endif
__device__ __forceinline__ void stream_store(u32x4 *p, const u32x4 v) {
#if SC_STREAM_STORES
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ u32x4 stream_load(const u32x4 *p) {
#if SC_STREAM_LOADS
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

// Wave-private LDS hand-off: LDS instructions of one wave execute in order, so
// only the compiler has to be kept from moving accesses across this point.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Inclusive scan over the 64 lanes of a wave under an associative (not necessarily commutative) operator:
// lane i ends with op(v[0], ..., v[i]) in lane order.
template <typename Op>
__device__ __forceinline__ uint64_t wave_scan_incl(uint64_t v, int lane, Op op) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t t = __shfl_up((unsigned long long)v, d, 64);
    if (lane >= d) v = op(t, v);
  }
  return v;
}

__device__ __forceinline__ uint32_t bits_of(uint32_t v) { return v ? 32u - (uint32_t)__builtin_clz(v) : 0u; }

}  // namespace sc
