// Development harness 4: one-shot decoder with DEDICATED prefetch workgroups.  Every group of P decode workgroups is
// preceded by one workgroup that only touches the payload (and seeds) of the group DIST groups ahead, so that the
// decode waves find their input in the memory-side cache instead of waiting on HBM behind the store traffic.
// The payload slice rotates through a 4 GiB region from launch to launch so that nothing stays cache-resident between reps.
#include "codec_kernels.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace sc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// P = decode workgroups per group, DIST = prefetch distance in groups (0 = no prefetch workgroups: baseline)
template <int B, int MODE, int U, int BS, int P, int DIST, int SPARSE>
__global__ __launch_bounds__(BS) void k_shot_pfd(UnpackArgs a, int dist) {
  __shared__ u32x4 lds[BS / 64][2][33];
  constexpr int PER_WG = (BS / 32) * U;        // blocks per decode workgroup
  constexpr int GROUP_BLOCKS = P * PER_WG;     // blocks per group
  const u32x4 *payload = reinterpret_cast<const u32x4 *>(a.payload);
  size_t dg = blockIdx.x;
  if constexpr (DIST > 0) {
    const size_t group = blockIdx.x / (P + 1);
    const int r = blockIdx.x % (P + 1);
    if (r == 0) {
      // prefetcher: touch the payload vectors [v0, v0 + GROUP_BLOCKS*B) of the group DIST ahead
      const size_t tgt = group + dist;
      const size_t lastv = a.nblocks * (size_t)B - 1;
      const size_t v0 = tgt * GROUP_BLOCKS * (size_t)B;
      constexpr int STEP = SPARSE ? 8 : 1;     // SPARSE: one 16-byte vector per 128-byte line
      constexpr int NV = GROUP_BLOCKS * B;
      constexpr int TRIPS = (NV + BS * STEP - 1) / (BS * STEP);
      u32x4 acc = {0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < TRIPS; ++t) {
        size_t v = v0 + ((size_t)t * BS + threadIdx.x) * STEP;
        v = v < lastv ? v : lastv;
        u32x4 x = payload[v];
        acc |= x;
      }
      if constexpr (MODE != MODE_PLAIN) {
        constexpr int SV = GROUP_BLOCKS / 4;   // seed vectors (4 seeds each)
        constexpr int STRIPS = (SV + BS * STEP - 1) / (BS * STEP);
        const u32x4 *sp = reinterpret_cast<const u32x4 *>(a.seeds);
        const size_t lasts = a.nblocks / 4 - 1;
#pragma unroll
        for (int t = 0; t < STRIPS; ++t) {
          size_t v = tgt * SV + ((size_t)t * BS + threadIdx.x) * STEP;
          v = v < lasts ? v : lasts;
          acc |= sp[v];
        }
      }
      asm volatile("" ::"v"(acc));
      return;
    }
    dg = group * P + (r - 1);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  u32x4 *slot = lds[wv][h];
  const int kk = k < B ? k : (B ? B - 1 : 0);
  const size_t first = (dg * (BS / 64) + wv) * 2 * U + h;
  u32x4 r[U];
  uint32_t sd[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t blk = first + 2 * (size_t)u;
    r[u] = payload[blk * (size_t)B + kk];
    if constexpr (MODE != MODE_PLAIN) sd[u] = a.seeds[blk];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t blk = first + 2 * (size_t)u;
    unpack_finish<B, MODE, 1>(r[u], B, MODE != MODE_PLAIN ? sd[u] : 0u, slot, k, a.out + blk * 128 + 4 * k);
  }
}

// XCD-aware variant: workgroup id % 8 is the XCD (and L2) a workgroup lands on.  A group is 8 prefetch workgroups
// followed by PG decode workgroups (PG % 8 == 0); prefetcher x touches the payload of the decode workgroups of the
// target group that will run on XCD x, so the lines wait in the L2 that will use them.
template <int B, int MODE, int U, int BS, int PG>
__global__ __launch_bounds__(BS) void k_shot_pfx(UnpackArgs a, uint32_t dist) {
  __shared__ u32x4 lds[BS / 64][2][33];
  constexpr int PER_WG = (BS / 32) * U;
  static_assert(PG % 8 == 0, "PG");
  const u32x4 *payload = reinterpret_cast<const u32x4 *>(a.payload);
  const uint32_t group = blockIdx.x / (PG + 8), r = blockIdx.x % (PG + 8);
  if (r < 8) {
    const size_t t = (size_t)group + dist;
    constexpr int VPW = PER_WG * B;            // payload vectors per decode workgroup
    constexpr int NV = (PG / 8) * VPW;         // vectors this prefetcher touches
    constexpr int TRIPS = (NV + BS - 1) / BS;
    const size_t lastv = a.nblocks * (size_t)B - 1;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int tr = 0; tr < TRIPS; ++tr) {
      const uint32_t f = tr * BS + threadIdx.x;
      const uint32_t i = f / VPW, within = f % VPW;
      size_t v = (t * PG + r + 8 * (size_t)i) * VPW + within;
      v = v < lastv ? v : lastv;
      acc |= payload[v];
    }
    if constexpr (MODE != MODE_PLAIN) {
      // seeds: PER_WG * 4 bytes per decode workgroup; one 16-byte vector covers 4 blocks
      constexpr int SVPW = PER_WG / 4;
      constexpr int SNV = (PG / 8) * SVPW;
      constexpr int STRIPS = (SNV + BS - 1) / BS;
      const u32x4 *sp = reinterpret_cast<const u32x4 *>(a.seeds);
      const size_t lasts = a.nblocks / 4 - 1;
#pragma unroll
      for (int tr = 0; tr < STRIPS; ++tr) {
        const uint32_t f = tr * BS + threadIdx.x;
        const uint32_t i = f / SVPW, within = f % SVPW;
        size_t v = (t * PG + r + 8 * (size_t)i) * SVPW + within;
        v = v < lasts ? v : lasts;
        acc |= sp[v];
      }
    }
    asm volatile("" ::"v"(acc));
    return;
  }
  const size_t dg = (size_t)group * PG + (r - 8);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  u32x4 *slot = lds[wv][h];
  const int kk = k < B ? k : (B ? B - 1 : 0);
  const size_t first = (dg * (BS / 64) + wv) * 2 * U + h;
  u32x4 rr[U];
  uint32_t sd[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t blk = first + 2 * (size_t)u;
    rr[u] = payload[blk * (size_t)B + kk];
    if constexpr (MODE != MODE_PLAIN) sd[u] = a.seeds[blk];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t blk = first + 2 * (size_t)u;
    unpack_finish<B, MODE, 1>(rr[u], B, MODE != MODE_PLAIN ? sd[u] : 0u, slot, k, a.out + blk * 128 + 4 * k);
  }
}

static size_t g_nblk;
static UnpackArgs g_a;
static uint8_t *g_region;
static size_t g_region_bytes;
static size_t g_cursor;
static uint32_t *g_seed_region;
static size_t g_seed_slices, g_seed_i;

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

template <int B, int U, int BS, int P, int DIST, int SPARSE = 0> static void run(int dist_mib = 0) {
  constexpr int PER_WG = (BS / 32) * U;
  size_t nblk = g_nblk / ((size_t)PER_WG * P) * ((size_t)PER_WG * P);
  size_t ngroups = nblk / ((size_t)PER_WG * P);
  size_t grid = DIST > 0 ? ngroups * (P + 1) : ngroups * P;
  size_t slice = (nblk * 16 * (size_t)B + 4095) / 4096 * 4096;
  int dist = (int)(((size_t)dist_mib << 20) / ((size_t)PER_WG * P * 512));
  if (dist < 1) dist = 1;
  double ms = time_ms([&] {
    UnpackArgs a = g_a; a.bit = B; a.nblocks = nblk;
    if (g_cursor + slice > g_region_bytes) g_cursor = 0;
    a.payload = g_region + g_cursor;
    g_cursor += slice;
    a.seeds = g_seed_region + (g_seed_i++ % g_seed_slices) * g_nblk;
    hipLaunchKernelGGL((k_shot_pfd<B, MODE_D1, U, BS, P, DIST, SPARSE>), dim3(grid), dim3(BS), 0, 0, a, dist);
  });
  double bytes = (double)nblk * (512 + 16 * B + 4);
  printf("b=%2d U=%d BS=%3d P=%4d pf=%d distMiB=%3d (D=%5d) sp=%d  %.4f ms %7.1f GB/s frac %.3f\n", B, U, BS, P, DIST, dist_mib, dist, SPARSE, ms,
         bytes / ms / 1e6, bytes / ms / 8e9);
  fflush(stdout);
}

template <int B, int U, int BS, int PG> static void runx(int dist_mib) {
  constexpr int PER_WG = (BS / 32) * U;
  size_t nblk = g_nblk / ((size_t)PER_WG * PG) * ((size_t)PER_WG * PG);
  size_t ngroups = nblk / ((size_t)PER_WG * PG);
  size_t grid = ngroups * (PG + 8);
  size_t slice = (nblk * 16 * (size_t)B + 4095) / 4096 * 4096;
  uint32_t dist = (uint32_t)(((size_t)dist_mib << 20) / ((size_t)PER_WG * PG * 512));
  if (dist < 1) dist = 1;
  double ms = time_ms([&] {
    UnpackArgs a = g_a; a.bit = B; a.nblocks = nblk;
    if (g_cursor + slice > g_region_bytes) g_cursor = 0;
    a.payload = g_region + g_cursor;
    g_cursor += slice;
    a.seeds = g_seed_region + (g_seed_i++ % g_seed_slices) * g_nblk;
    hipLaunchKernelGGL((k_shot_pfx<B, MODE_D1, U, BS, PG>), dim3(grid), dim3(BS), 0, 0, a, dist);
  });
  double bytes = (double)nblk * (512 + 16 * B + 4);
  printf("b=%2d xcd-aware PG=%3d distMiB=%2d (D=%3u)  %.4f ms %7.1f GB/s frac %.3f\n", B, PG, dist_mib, dist, ms, bytes / ms / 1e6, bytes / ms / 8e9);
  fflush(stdout);
}
template <int B> static void prod() {
  size_t nblk = g_nblk;
  size_t slice = (nblk * 16 * (size_t)B + 4095) / 4096 * 4096;
  double ms = time_ms([&] {
    UnpackArgs a = g_a; a.bit = B; a.nblocks = nblk;
    if (g_cursor + slice > g_region_bytes) g_cursor = 0;
    a.payload = g_region + g_cursor;
    g_cursor += slice;
    a.seeds = g_seed_region + (g_seed_i++ % g_seed_slices) * g_nblk;
    (void)launch_unpack(MODE_D1, false, a, 0);
  });
  double bytes = (double)nblk * (512 + 16 * B + 4);
  printf("b=%2d production  %.4f ms %7.1f GB/s frac %.3f\n", B, ms, bytes / ms / 1e6, bytes / ms / 8e9);
}
template <int B, int U, int BS> static void sweep() {
  prod<B>();
  for (int mib : {2, 4, 8, 12, 20, 32}) runx<B, U, BS, 128>(mib);
  for (int mib : {2, 4, 8, 12, 20, 32}) runx<B, U, BS, 64>(mib);
}

int main(int argc, char **argv) {
  int log2n = argc > 1 ? atoi(argv[1]) : 30;
  size_t n = (size_t)1 << log2n;
  g_nblk = n / 128;
  uint32_t *out, *seeds;
  g_region_bytes = (size_t)4 << 30;
  CK(hipMalloc(&out, n * 4)); g_seed_slices = 24; CK(hipMalloc(&g_seed_region, g_seed_slices * g_nblk * 4)); seeds = g_seed_region; CK(hipMalloc(&g_region, g_region_bytes + 4096));
  CK(hipMemset(g_region, 0x5b, g_region_bytes)); CK(hipMemset(seeds, 1, g_seed_slices * g_nblk * 4));
  g_a = UnpackArgs{}; g_a.seeds = seeds; g_a.out = out; g_a.nblocks = g_nblk; g_a.length = n;
  sweep<1, 2, 128>(); sweep<2, 2, 128>(); sweep<4, 2, 128>(); sweep<7, 2, 128>(); sweep<10, 2, 128>();
  return 0;
}
