// Development harness: one-shot decoder, workgroup size x steps-per-wave sweep.
#include "codec_kernels.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace sc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int B, int MODE, int U, int BS, int STORE = 0, int STRIDE = B, int NVEC = B>
__global__ __launch_bounds__(BS) void k_shot(UnpackArgs a) {
  __shared__ u32x4 lds[BS / 64][2][33];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  u32x4 *slot = lds[wv][h];
  const u32x4 *payload = reinterpret_cast<const u32x4 *>(a.payload);
  const int kk = k < NVEC ? k : (NVEC ? NVEC - 1 : 0);
  const size_t first = ((size_t)blockIdx.x * (BS / 64) + wv) * 2 * U + h;
  u32x4 r[U];
  uint32_t sd[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t blk = first + 2 * (size_t)u;
    if constexpr (B != 0) r[u] = payload[blk * (size_t)STRIDE + kk];
    if constexpr (MODE != MODE_PLAIN) sd[u] = a.seeds[blk];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t blk = first + 2 * (size_t)u;
    unpack_finish<B, MODE, STORE>(r[u], B, MODE != MODE_PLAIN ? sd[u] : 0u, slot, k, a.out + blk * 128 + 4 * k);
  }
}

// same kernel with its hot pointers as separate scalar arguments (candidates for kernarg preload into SGPRs)
template <int B, int MODE, int U, int BS>
__global__ __launch_bounds__(BS) void k_shot_args(const u32x4 *payload, const uint32_t *seeds, uint32_t *out) {
  __shared__ u32x4 lds[BS / 64][2][33];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  u32x4 *slot = lds[wv][h];
  const int kk = k < B ? k : (B ? B - 1 : 0);
  const size_t first = ((size_t)blockIdx.x * (BS / 64) + wv) * 2 * U + h;
  u32x4 r[U];
  uint32_t sd[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t blk = first + 2 * (size_t)u;
    r[u] = payload[blk * (size_t)B + kk];
    sd[u] = seeds[blk];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t blk = first + 2 * (size_t)u;
    unpack_finish<B, MODE, 1>(r[u], B, sd[u], slot, k, out + blk * 128 + 4 * k);
  }
}

template <typename F>
static double time_ms(F f, int reps = 9) {
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
static UnpackArgs g_a;

template <int B, int U, int BS, int STORE = 0, int STRIDE = B, int NVEC = B> static void run() {
  UnpackArgs a = g_a; a.bit = B;
  size_t grid = g_nblk / ((BS / 64) * 2 * U);
  double ms = time_ms([&] { hipLaunchKernelGGL((k_shot<B, MODE_D1, U, BS, STORE, STRIDE, NVEC>), dim3(grid), dim3(BS), 0, 0, a); });
  double bytes = (double)g_nblk * (512 + 16 * B + 4);
  printf("b=%2d U=%d BS=%4d st=%d stride=%d nvec=%d  %.4f ms %7.1f GB/s frac %.3f\n", B, U, BS, STORE, STRIDE, NVEC, ms, bytes / ms / 1e6, bytes / ms / 8e9);
}
template <int B, int U, int BS> static void run_args() {
  size_t grid = g_nblk / ((BS / 64) * 2 * U);
  double ms = time_ms([&] { hipLaunchKernelGGL((k_shot_args<B, MODE_D1, U, BS>), dim3(grid), dim3(BS), 0, 0, (const u32x4 *)g_a.payload, g_a.seeds, g_a.out); });
  double bytes = (double)g_nblk * (512 + 16 * B + 4);
  printf("b=%2d U=%d BS=%4d args        %.4f ms %7.1f GB/s frac %.3f\n", B, U, BS, ms, bytes / ms / 1e6, bytes / ms / 8e9);
}
template <int B> static void sweep() {
  run<B, 2, 128, 1>(); run_args<B, 2, 128>(); run<B, 1, 128, 1>(); run_args<B, 1, 128>();
}
template <int... Bs> static void sweep_all(std::integer_sequence<int, Bs...>) { (sweep<Bs + 1>(), ...); }
int main(int argc, char **argv) {
  int log2n = argc > 1 ? atoi(argv[1]) : 28;
  size_t n = (size_t)1 << log2n;
  g_nblk = n / 128;
  // 3 * 2^k blocks so that every U in {1,2,3,4} divides evenly
  g_nblk = g_nblk / 3072 * 3072;
  uint32_t *out, *seeds; uint8_t *payload;
  CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&seeds, g_nblk * 4)); CK(hipMalloc(&payload, g_nblk * 512 + 4096));
  CK(hipMemset(payload, 0x5b, g_nblk * 512)); CK(hipMemset(seeds, 1, g_nblk * 4));
  g_a = UnpackArgs{}; g_a.payload = payload; g_a.seeds = seeds; g_a.out = out; g_a.nblocks = g_nblk; g_a.length = n;
  sweep<2>(); sweep<3>(); sweep<4>(); sweep<8>(); sweep<16>(); sweep<24>();
  return 0;
}
