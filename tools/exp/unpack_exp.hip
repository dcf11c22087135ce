// Development harness: times kernel variants of the decoder next to plain fill /
// copy kernels on one GPU.  Build: hipcc -O3 --offload-arch=gfx950 -I simdcomp_amd/csrc tools/exp/unpack_exp.hip -o /tmp/unpack_exp
#include "codec_kernels.hip"
#include <cstdio>
#include <vector>
#include <algorithm>

using namespace sc;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_fill(u32x4 *out, size_t nvec, uint32_t val) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) out[i] = u32x4{val, val, val, val};
}
__global__ __launch_bounds__(256) void k_fill_nt(u32x4 *out, size_t nvec, uint32_t val) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(u32x4{val, val, val, val}, &out[i]);
}
__global__ __launch_bounds__(256) void k_copy(const u32x4 *in, u32x4 *out, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
template <int U>
__global__ __launch_bounds__(256) void k_copy_u(const u32x4 *in, u32x4 *out, size_t nvec) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < nvec; i += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) out[i + u * stride] = v[u];
  }
}

// one shot: no loop, every wave decodes 2*U consecutive blocks and ends
template <int B, int MODE, int U, int BS>
__global__ __launch_bounds__(BS) void k_unpack_shot(UnpackArgs a) {
  __shared__ u32x4 lds[BS / 64][2][33];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  u32x4 *slot = lds[wv][h];
  const u32x4 *payload = reinterpret_cast<const u32x4 *>(a.payload);
  const int kk = k < B ? k : (B ? B - 1 : 0);
  const size_t first = ((size_t)blockIdx.x * (BS / 64) + wv) * 2 * U + h;
  u32x4 r[U];
  uint32_t sd[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t blk = first + 2 * (size_t)u;
    if constexpr (B != 0) r[u] = payload[blk * (size_t)B + kk];
    if constexpr (MODE != MODE_PLAIN) sd[u] = a.seeds[blk];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t blk = first + 2 * (size_t)u;
    unpack_finish<B, MODE>(r[u], B, MODE != MODE_PLAIN ? sd[u] : 0u, slot, k, a.out + blk * 128 + 4 * k);
  }
}

// one shot with an L2 prefetch of the payload lines (and seeds) that the workgroup DIST ahead will need,
// issued behind the wave's own loads so that its latency overlaps the store acknowledgement
template <int B, int MODE, int BS, int DIST>
__global__ __launch_bounds__(BS) void k_unpack_shot_pf(UnpackArgs a) {
  __shared__ u32x4 lds[BS / 64][2][33];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  u32x4 *slot = lds[wv][h];
  const u32x4 *payload = reinterpret_cast<const u32x4 *>(a.payload);
  const int kk = k < B ? k : (B ? B - 1 : 0);
  constexpr int BPW = (BS / 64) * 2;  // blocks per workgroup
  const size_t blk = ((size_t)blockIdx.x * (BS / 64) + wv) * 2 + h;
  u32x4 r;
  uint32_t sd = 0;
  if constexpr (B != 0) r = payload[blk * (size_t)B + kk];
  if constexpr (MODE != MODE_PLAIN) sd = a.seeds[blk];
  // prefetch target: workgroup blockIdx.x + DIST (same XCD when DIST is a multiple of 8), clamped to the grid
  size_t pwg = (size_t)blockIdx.x + DIST;
  if (pwg >= gridDim.x) pwg = blockIdx.x;
  // B payload lines of 128 bytes per 8 blocks (BS=256); lanes 0..lines-1 of wave 0 touch one dword per line, lane 63 the seeds
  constexpr int LINES = (BPW * 16 * B + 127) / 128;
  const char *pf = reinterpret_cast<const char *>(payload) + pwg * (size_t)(BPW * 16 * B) + (size_t)lane * 128;
  if (MODE != MODE_PLAIN && lane == 63) pf = reinterpret_cast<const char *>(a.seeds + pwg * BPW);
  const bool do_pf = wv == 0 && (lane < LINES || (MODE != MODE_PLAIN && lane == 63));
  uint32_t dummy = 0;
  // compute first (all compiler-visible waits happen here) ...
  u32x4 v;
  {
    if constexpr (B == 0) v = u32x4{0, 0, 0, 0};
    else if constexpr (B == 32) v = r;
    else {
      wave_lds_fence(); slot[k] = r; wave_lds_fence();
      const uint32_t p = (uint32_t)k * B, w = p >> 5, s = p & 31;
      const u32x4 lo = slot[w], hi = slot[w + 1];
      const uint32_t m = field_mask(B);
      v.x = funnel(hi.x, lo.x, s) & m; v.y = funnel(hi.y, lo.y, s) & m; v.z = funnel(hi.z, lo.z, s) & m; v.w = funnel(hi.w, lo.w, s) & m;
    }
    if constexpr (MODE == MODE_D1 && B != 32) {
      v.y += v.x; v.z += v.y; v.w += v.z;
      const uint32_t incl = half_scan_add(v.w);
      const uint32_t before = sd + incl - v.w;
      v.x += before; v.y += before; v.z += before; v.w += before;
    } else if constexpr (MODE == MODE_FOR && B != 32) { v.x += sd; v.y += sd; v.z += sd; v.w += sd; }
  }
  if (do_pf) asm volatile("global_load_dword %0, %1, off" : "=v"(dummy) : "v"(pf) : "memory");
  *reinterpret_cast<u32x4 *>(a.out + blk * 128 + 4 * k) = v;
  asm volatile("" ::"v"(dummy));
}

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
static UnpackArgs g_a;

template <int B, int MODE, int U, bool PIPE>
static void run_variant(const char *name, int wg_per_cu) {
  UnpackArgs a = g_a; a.bit = B;
  const size_t per_iter = 8 * U, ntiles = g_nblk / per_iter;
  size_t grid = std::min<size_t>(ntiles, (size_t)256 * wg_per_cu);
  double ms = time_ms([&] {
    if (PIPE) hipLaunchKernelGGL((k_unpack_pipe<B, MODE, U>), dim3(grid), dim3(256), 0, 0, a, ntiles);
    else hipLaunchKernelGGL((k_unpack_fast<B, MODE, U>), dim3(grid), dim3(256), 0, 0, a, ntiles);
  });
  double bytes = (double)g_nblk * (512 + 16 * B + (MODE ? 4 : 0));
  printf("%-28s b=%2d U=%d wg/cu=%2d  %.4f ms  %7.1f GB/s  frac %.3f\n", name, B, U, wg_per_cu, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
}

template <int B, int MODE, int U, int BS>
static void run_shot(const char *name) {
  UnpackArgs a = g_a; a.bit = B;
  const size_t per_wg = (BS / 64) * 2 * U;
  size_t grid = g_nblk / per_wg;
  double ms = time_ms([&] { hipLaunchKernelGGL((k_unpack_shot<B, MODE, U, BS>), dim3(grid), dim3(BS), 0, 0, a); });
  double bytes = (double)g_nblk * (512 + 16 * B + (MODE ? 4 : 0));
  printf("%-28s b=%2d U=%d BS=%4d     %.4f ms  %7.1f GB/s  frac %.3f\n", name, B, U, BS, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
}

template <int B, int MODE, int BS, int DIST>
static void run_shot_pf(const char *name) {
  UnpackArgs a = g_a; a.bit = B;
  const size_t per_wg = (BS / 64) * 2;
  size_t grid = g_nblk / per_wg;
  double ms = time_ms([&] { hipLaunchKernelGGL((k_unpack_shot_pf<B, MODE, BS, DIST>), dim3(grid), dim3(BS), 0, 0, a); });
  double bytes = (double)g_nblk * (512 + 16 * B + (MODE ? 4 : 0));
  printf("%-28s b=%2d D=%5d BS=%4d  %.4f ms  %7.1f GB/s  frac %.3f\n", name, B, DIST, BS, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
}

template <int B>
static void sweep_b() {
  run_shot_pf<B, MODE_D1, 256, 1024>("shot_pf d1");
  run_shot_pf<B, MODE_D1, 256, 2048>("shot_pf d1");
  run_shot_pf<B, MODE_D1, 256, 4096>("shot_pf d1");
  run_shot_pf<B, MODE_D1, 256, 8192>("shot_pf d1");
  run_shot_pf<B, MODE_D1, 256, 16384>("shot_pf d1");
  run_shot_pf<B, MODE_D1, 256, 4099>("shot_pf d1");
  run_shot_pf<B, MODE_PLAIN, 256, 4096>("shot_pf plain");
  run_shot<B, MODE_D1, 1, 256>("shot d1");
  run_shot<B, MODE_D1, 2, 256>("shot d1");
  run_shot<B, MODE_PLAIN, 1, 256>("shot plain");
  run_shot<B, MODE_FOR, 1, 256>("shot for");
  run_variant<B, MODE_D1, 4, false>("fast d1", 8);
  run_variant<B, MODE_D1, 4, true>("pipe d1", 8);
}

int main(int argc, char **argv) {
  int log2n = argc > 1 ? atoi(argv[1]) : 28;
  size_t n = (size_t)1 << log2n;
  g_nblk = n / 128;
  uint32_t *out, *seeds; uint8_t *payload;
  CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&seeds, g_nblk * 4)); CK(hipMalloc(&payload, g_nblk * 512 + 4096));
  CK(hipMemset(payload, 0x5b, g_nblk * 512)); CK(hipMemset(seeds, 1, g_nblk * 4));
  g_a = UnpackArgs{}; g_a.payload = payload; g_a.seeds = seeds; g_a.out = out; g_a.nblocks = g_nblk; g_a.length = n;
  size_t nvec = n / 4;
  for (int g : {2048}) {
    double ms = time_ms([&] { hipLaunchKernelGGL(k_fill, dim3(g), dim3(256), 0, 0, (u32x4 *)out, nvec, 7u); });
    printf("fill      grid=%5d  %.4f ms %7.1f GB/s\n", g, ms, n * 4.0 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_fill_nt, dim3(g), dim3(256), 0, 0, (u32x4 *)out, nvec, 7u); });
    printf("fill_nt   grid=%5d  %.4f ms %7.1f GB/s\n", g, ms, n * 4.0 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, (const u32x4 *)payload, (u32x4 *)out, nvec); });
    printf("copy      grid=%5d  %.4f ms %7.1f GB/s\n", g, ms, n * 8.0 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_copy_u<4>, dim3(g), dim3(256), 0, 0, (const u32x4 *)payload, (u32x4 *)out, nvec); });
    printf("copy_u4   grid=%5d  %.4f ms %7.1f GB/s\n", g, ms, n * 8.0 / ms / 1e6);
  }
  sweep_b<1>(); sweep_b<2>(); sweep_b<4>(); sweep_b<8>(); sweep_b<16>(); sweep_b<31>(); sweep_b<32>();
  return 0;
}
