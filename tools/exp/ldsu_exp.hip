// Probe: 16-byte LDS reads at arbitrary byte addresses (correctness + rate against aligned reads), and the latency of
// a chain of dependent LDS byte reads (the record walk of the byte stream).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

template <int MODE>  // 0: aligned 16, 1: any byte, 2: 4-byte aligned
__global__ __launch_bounds__(256) void k_read(const uint8_t *src, uint32_t *out, int shift, int iters) {
  __shared__ u32x4 sh[1024 + 2];
  for (int i = threadIdx.x; i < 1026; i += 256) sh[i] = reinterpret_cast<const u32x4 *>(src)[i];
  __syncthreads();
  const uint8_t *by = reinterpret_cast<const uint8_t *>(sh);
  u32x4 acc = {0, 0, 0, 0};
  uint32_t at = threadIdx.x * 16 + shift;
  for (int it = 0; it < iters; ++it) {
    u32x4 v;
    if (MODE == 0) v = *reinterpret_cast<const u32x4 *>(by + (at & ~15u));
    else if (MODE == 1) v = *reinterpret_cast<const u32x4_u *>(by + at);
    else v = *reinterpret_cast<const u32x4_a4 *>(by + (at & ~3u));
    acc += v;
    at = (at + 4096 + 16) & 16383;
  }
  reinterpret_cast<u32x4 *>(out)[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_check(const uint8_t *src, uint8_t *out, int shift) {
  __shared__ u32x4 sh[1024 + 2];
  for (int i = threadIdx.x; i < 1026; i += 256) sh[i] = reinterpret_cast<const u32x4 *>(src)[i];
  __syncthreads();
  const uint8_t *by = reinterpret_cast<const uint8_t *>(sh);
  for (int i = threadIdx.x; i < 1024; i += 256) {
    u32x4 v = *reinterpret_cast<const u32x4_u *>(by + 16 * i + shift);
    reinterpret_cast<u32x4 *>(out)[i] = v;
  }
}

// one lane per wave walks: p += 1 + 16 * (byte[p] & 3)
__global__ __launch_bounds__(256) void k_walk(const uint8_t *src, uint32_t *out, long long *cycles) {
  __shared__ u32x4 sh[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) sh[i] = reinterpret_cast<const u32x4 *>(src)[i];
  __syncthreads();
  const uint8_t *by = reinterpret_cast<const uint8_t *>(sh);
  uint32_t p = threadIdx.x & 7, n = 0;
  long long t0 = clock64();
  while (p < 16384) {
    p += 1 + 16 * (by[p] & 7);
    ++n;
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { cycles[0] = t1 - t0; cycles[1] = n; }
  out[blockIdx.x * 256 + threadIdx.x] = n + p;
}

int main() {
  std::vector<uint8_t> h(16 * 1026), o(16 * 1024);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 131 + (i >> 8) * 7);
  uint8_t *s, *d;
  uint32_t *out;
  long long *cyc;
  CK(hipMalloc(&s, h.size())); CK(hipMalloc(&d, o.size())); CK(hipMalloc(&out, 16 * 256 * 4096)); CK(hipMalloc(&cyc, 16));
  CK(hipMemcpy(s, h.data(), h.size(), hipMemcpyHostToDevice));
  int bad = 0;
  for (int shift = 0; shift < 16; ++shift) {
    hipLaunchKernelGGL(k_check, dim3(1), dim3(256), 0, 0, s, d, shift);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(o.data(), d, o.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < o.size(); ++i) if (o[i] != h[i + shift]) { ++bad; break; }
  }
  printf("unaligned LDS 16-byte reads: %s (%d shifts bad)\n", bad ? "WRONG" : "ok", bad);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2048;
  for (int mode = 0; mode < 3; ++mode) for (int shift : {0, 4, 1, 7}) {
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k_read<0>, dim3(4096), dim3(256), 0, 0, s, out, shift, iters);
      if (mode == 1) hipLaunchKernelGGL(k_read<1>, dim3(4096), dim3(256), 0, 0, s, out, shift, iters);
      if (mode == 2) hipLaunchKernelGGL(k_read<2>, dim3(4096), dim3(256), 0, 0, s, out, shift, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r && ms < best) best = ms;
    }
    printf("mode %d shift %d: %.3f ms  %.1f TB/s of LDS reads\n", mode, shift, best, 4096.0 * 256 * iters * 16 / best / 1e9);
  }
  hipLaunchKernelGGL(k_walk, dim3(1), dim3(256), 0, 0, s, out, cyc);
  CK(hipDeviceSynchronize());
  long long hc[2];
  CK(hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost));
  printf("walk: %lld steps, %lld clock64 ticks, %.1f per step (clock64 runs at 100 MHz)\n", hc[1], hc[0], (double)hc[0] / hc[1]);
  return bad != 0;
}
