// Development harness: one-shot encoder, workgroup size x steps sweep (d1, unmasked).
#include "codec_kernels.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace sc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <typename F>
static double time_ms(F f, int reps = 7) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}
static size_t g_nblk;
static PackArgs g_a;
template <int B, int U, int BS> static void run() {
  PackArgs a = g_a; a.bit = B;
  size_t grid = g_nblk / ((BS / 32) * U);
  double ms = time_ms([&] { hipLaunchKernelGGL((k_pack_fast<B, MODE_D1, false, U, BS>), dim3(grid), dim3(BS), 0, 0, a); });
  double bytes = (double)g_nblk * (512 + 16 * B + 4);
  printf("b=%2d U=%d BS=%4d  %.4f ms %7.1f GB/s frac %.3f\n", B, U, BS, ms, bytes / ms / 1e6, bytes / ms / 8e9);
}
template <int B> static void sweep() {
  run<B, 1, 128>(); run<B, 2, 128>(); run<B, 4, 128>(); run<B, 1, 256>(); run<B, 2, 256>(); run<B, 4, 256>(); run<B, 1, 512>();
}
template <int... Bs> static void sweep_all(std::integer_sequence<int, Bs...>) { (sweep<Bs + 1>(), ...); }
int main(int argc, char **argv) {
  int log2n = argc > 1 ? atoi(argv[1]) : 30;
  size_t n = (size_t)1 << log2n;
  g_nblk = n / 128;
  uint32_t *in, *seeds; uint8_t *payload;
  CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&seeds, g_nblk * 4)); CK(hipMalloc(&payload, g_nblk * 512 + 4096));
  CK(hipMemset(in, 0x5b, n * 4)); CK(hipMemset(seeds, 1, g_nblk * 4));
  g_a = PackArgs{}; g_a.in = in; g_a.payload = payload; g_a.seeds = seeds; g_a.nblocks = g_nblk; g_a.length = n;
  sweep_all(std::make_integer_sequence<int, 32>{});
  return 0;
}
