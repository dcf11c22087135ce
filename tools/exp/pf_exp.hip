// Development harness 3: one-shot decoder with an asm-issued prefetch for the workgroup DIST ahead
// (own loads + prefetch loads in one asm block, s_waitcnt leaves the prefetch in flight).
#include "codec_kernels.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace sc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int B, int MODE, int U, int DIST>
__global__ __launch_bounds__(256) void k_shot_pf(UnpackArgs a) {
  __shared__ u32x4 lds[4][2][33];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  u32x4 *slot = lds[wv][h];
  const u32x4 *payload = reinterpret_cast<const u32x4 *>(a.payload);
  const int kk = k < B ? k : (B ? B - 1 : 0);
  constexpr size_t BPW = 8 * U;
  const size_t first = ((size_t)blockIdx.x * 4 + wv) * 2 * U + h;
  size_t pwg = (size_t)blockIdx.x + DIST;
  const bool ahead = pwg < gridDim.x;
  const size_t pfirst = first + (ahead ? (size_t)DIST * BPW : 0);
  u32x4 r[U];
  uint32_t sd[U];
  u32x4 dump;
  uint32_t dump2;
  static_assert(U == 1 || U == 2, "U");
  if constexpr (U == 1) {
    const u32x4 *p0 = payload + first * (size_t)B + kk;
    const uint32_t *s0 = a.seeds + first;
    const u32x4 *q0 = payload + pfirst * (size_t)B + kk;
    const uint32_t *t0 = a.seeds + pfirst;
    asm volatile(
        "global_load_dwordx4 %0, %4, off\n\t"
        "global_load_dword %1, %5, off\n\t"
        "global_load_dwordx4 %2, %6, off\n\t"
        "global_load_dword %3, %7, off\n\t"
        "s_waitcnt vmcnt(2)"
        : "=&v"(r[0]), "=&v"(sd[0]), "=&v"(dump), "=&v"(dump2)
        : "v"(p0), "v"(s0), "v"(q0), "v"(t0)
        : "memory");
  } else {
    const u32x4 *p0 = payload + first * (size_t)B + kk, *p1 = payload + (first + 2) * (size_t)B + kk;
    const uint32_t *s0 = a.seeds + first;
    const u32x4 *q0 = payload + pfirst * (size_t)B + kk, *q1 = payload + (pfirst + 2) * (size_t)B + kk;
    const uint32_t *t0 = a.seeds + pfirst;
    u32x4 dumpb;
    asm volatile(
        "global_load_dwordx4 %0, %7, off\n\t"
        "global_load_dwordx4 %1, %8, off\n\t"
        "global_load_dword %2, %9, off\n\t"
        "global_load_dword %3, %9, off offset:8\n\t"
        "global_load_dwordx4 %4, %10, off\n\t"
        "global_load_dwordx4 %5, %11, off\n\t"
        "global_load_dword %6, %12, off\n\t"
        "s_waitcnt vmcnt(3)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(sd[0]), "=&v"(sd[1]), "=&v"(dump), "=&v"(dumpb), "=&v"(dump2)
        : "v"(p0), "v"(p1), "v"(s0), "v"(q0), "v"(q1), "v"(t0)
        : "memory");
    asm volatile("" ::"v"(dumpb));
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t blk = first + 2 * (size_t)u;
    unpack_finish<B, MODE, 1>(r[u], B, sd[u], slot, k, a.out + blk * 128 + 4 * k);
  }
  asm volatile("" ::"v"(dump), "v"(dump2));
}

template <int B, int MODE, int U>
__global__ __launch_bounds__(256) void k_shot_nt(UnpackArgs a) {
  __shared__ u32x4 lds[4][2][33];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  u32x4 *slot = lds[wv][h];
  const u32x4 *payload = reinterpret_cast<const u32x4 *>(a.payload);
  const int kk = k < B ? k : (B ? B - 1 : 0);
  const size_t first = ((size_t)blockIdx.x * 4 + wv) * 2 * U + h;
  u32x4 r[U];
  uint32_t sd[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t blk = first + 2 * (size_t)u;
    r[u] = payload[blk * (size_t)B + kk];
    sd[u] = a.seeds[blk];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t blk = first + 2 * (size_t)u;
    unpack_finish<B, MODE, 1>(r[u], B, sd[u], slot, k, a.out + blk * 128 + 4 * k);
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
static uint32_t *g_ref;

template <int B, int U, int DIST> static void run_pf() {
  UnpackArgs a = g_a; a.bit = B;
  size_t grid = g_nblk / (8 * U);
  double ms = time_ms([&] { hipLaunchKernelGGL((k_shot_pf<B, MODE_D1, U, DIST>), dim3(grid), dim3(256), 0, 0, a); });
  double bytes = (double)g_nblk * (512 + 16 * B + 4);
  printf("pf   b=%2d U=%d D=%6d  %.4f ms %7.1f GB/s frac %.3f\n", B, U, DIST, ms, bytes / ms / 1e6, bytes / ms / 8e9);
}
template <int B, int U> static void run_base() {
  UnpackArgs a = g_a; a.bit = B;
  size_t grid = g_nblk / (8 * U);
  double ms = time_ms([&] { hipLaunchKernelGGL((k_shot_nt<B, MODE_D1, U>), dim3(grid), dim3(256), 0, 0, a); });
  double bytes = (double)g_nblk * (512 + 16 * B + 4);
  printf("base b=%2d U=%d           %.4f ms %7.1f GB/s frac %.3f\n", B, U, ms, bytes / ms / 1e6, bytes / ms / 8e9);
}
template <int B> static void sweep() {
  run_base<B, 1>(); run_base<B, 2>();
  run_pf<B, 1, 1024>(); run_pf<B, 1, 4096>(); run_pf<B, 1, 16384>(); run_pf<B, 1, 65536>(); run_pf<B, 1, 262144>();
  run_pf<B, 2, 1024>(); run_pf<B, 2, 4096>(); run_pf<B, 2, 16384>(); run_pf<B, 2, 65536>();
}
int main(int argc, char **argv) {
  int log2n = argc > 1 ? atoi(argv[1]) : 28;
  size_t n = (size_t)1 << log2n;
  g_nblk = n / 128;
  uint32_t *out, *seeds; uint8_t *payload;
  CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&seeds, g_nblk * 4)); CK(hipMalloc(&payload, g_nblk * 512 + 4096));
  CK(hipMemset(payload, 0x5b, g_nblk * 512)); CK(hipMemset(seeds, 1, g_nblk * 4));
  g_a = UnpackArgs{}; g_a.payload = payload; g_a.seeds = seeds; g_a.out = out; g_a.nblocks = g_nblk; g_a.length = n;
  sweep<2>(); sweep<4>(); sweep<8>(); sweep<16>();
  return 0;
}
