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

constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142;
constexpr int DPP_WAVE_SHR1 = 0x138;

// Lane l receives `v` from the lane the DPP control names, or 0 where that lane
// does not exist / the row is masked off.
template <int CTRL, int ROWMASK = 0xF>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xF, false);
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

__device__ __forceinline__ uint32_t bits_of(uint32_t v) { return v ? 32u - (uint32_t)__builtin_clz(v) : 0u; }

}  // namespace sc
