#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
__global__ void k_copy(const uint8_t* src, uint8_t* dst, int shift_in, int shift_out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32x4 v = *reinterpret_cast<const u32x4_u*>(src + shift_in + 16 * i);
  *reinterpret_cast<u32x4_u*>(dst + shift_out + 16 * i) = v;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
int main() {
  const int n = 1 << 16;
  std::vector<uint8_t> h(16 * n + 64), o(16 * n + 64);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 131 + (i >> 8));
  uint8_t *s, *d;
  CK(hipMalloc(&s, h.size())); CK(hipMalloc(&d, h.size()));
  CK(hipMemcpy(s, h.data(), h.size(), hipMemcpyHostToDevice));
  int bad = 0;
  for (int si = 0; si < 16; si += 1) for (int so = 0; so < 16; so += 5) {
    CK(hipMemset(d, 0xEE, h.size()));
    hipLaunchKernelGGL(k_copy, dim3(n / 256), dim3(256), 0, 0, s, d, si, so, n);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(o.data(), d, h.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < 16 * n; ++i) if (o[so + i] != h[si + i]) { ++bad; break; }
    for (int i = 0; i < so; ++i) if (o[i] != 0xEE) { ++bad; break; }
    if (o[so + 16 * n] != 0xEE) ++bad;
  }
  printf("unaligned dwordx4 copy: %s (%d bad)\n", bad ? "WRONG" : "ok", bad);
  // rate of a 1 GiB copy with the load side, the store side, or neither at an odd byte address
  const size_t big = (size_t)1 << 30;
  uint8_t *bs, *bd;
  CK(hipMalloc(&bs, big + 64)); CK(hipMalloc(&bd, big + 64));
  CK(hipMemset(bs, 1, big + 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int shifts[4][2] = {{0, 0}, {1, 0}, {0, 1}, {7, 3}};
  for (auto &sh : shifts) {
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_copy, dim3((unsigned)(big / 16 / 256)), dim3(256), 0, 0, bs, bd, sh[0], sh[1], (int)(big / 16));
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r && ms < best) best = ms;
    }
    printf("copy 1 GiB, load shift %d store shift %d: %.3f ms  %.0f GB/s\n", sh[0], sh[1], best, 2.0 * big / best / 1e6);
  }
  return bad != 0;
}
