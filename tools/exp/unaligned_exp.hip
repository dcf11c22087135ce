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
  return bad != 0;
}
