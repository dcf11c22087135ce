// Development harness 2: store-pattern ceilings with per-wave contiguous chunks and a dependent tiny load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// one shot, every wave writes CH KiB contiguous (CH stores of 1 KiB)
template <int CH, int BS>
__global__ __launch_bounds__(BS) void fill_wave(u32x4 *out, uint32_t val) {
  const size_t wave = (size_t)blockIdx.x * (BS / 64) + (threadIdx.x >> 6);
  u32x4 *o = out + wave * 64 * CH + (threadIdx.x & 63);
#pragma unroll
  for (int j = 0; j < CH; ++j) o[j * 64] = u32x4{val, val, val, val};
}
template <int CH, int BS>
__global__ __launch_bounds__(BS) void fill_wave_nt(u32x4 *out, uint32_t val) {
  const size_t wave = (size_t)blockIdx.x * (BS / 64) + (threadIdx.x >> 6);
  u32x4 *o = out + wave * 64 * CH + (threadIdx.x & 63);
#pragma unroll
  for (int j = 0; j < CH; ++j) __builtin_nontemporal_store(u32x4{val, val, val, val}, &o[j * 64]);
}
template <int CH, bool DRAIN>
__global__ __launch_bounds__(256) void fill_wave_gs_nt(u32x4 *out, size_t nwavechunks, uint32_t val) {
  for (size_t c = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); c < nwavechunks; c += (size_t)gridDim.x * 4) {
    u32x4 *o = out + c * 64 * CH + (threadIdx.x & 63);
#pragma unroll
    for (int j = 0; j < CH; ++j) __builtin_nontemporal_store(u32x4{val, val, val, val}, &o[j * 64]);
    if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
// same with a dependent load: DEP=1 -> 16 B per half-wave at "payload of block" address (b=1 pattern, HBM),
// DEP=2 -> everyone reads the same 1 KiB (cache resident)
template <int CH, int BS, int DEP>
__global__ __launch_bounds__(BS) void fill_wave_dep(const u32x4 *in, u32x4 *out) {
  const size_t wave = (size_t)blockIdx.x * (BS / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  u32x4 v[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const size_t blk = (wave * CH + j) * 2 + (lane >> 5);
    v[j] = DEP == 1 ? in[blk] : in[lane & 1];
  }
  u32x4 *o = out + wave * 64 * CH + lane;
#pragma unroll
  for (int j = 0; j < CH; ++j) o[j * 64] = v[j];
}
// persistent, wave-contiguous, optional drain of the stores every trip
template <int CH, bool DRAIN>
__global__ __launch_bounds__(256) void fill_wave_gs(u32x4 *out, size_t nwavechunks, uint32_t val) {
  for (size_t c = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); c < nwavechunks; c += (size_t)gridDim.x * 4) {
    u32x4 *o = out + c * 64 * CH + (threadIdx.x & 63);
#pragma unroll
    for (int j = 0; j < CH; ++j) o[j * 64] = u32x4{val, val, val, val};
    if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

static size_t nvec;
static u32x4 *out, *in;
static double gb;
#define RUN(name, ...) { double ms = time_ms([&] { __VA_ARGS__; }); printf("%-40s %.4f ms %8.1f GB/s\n", name, ms, gb / ms); }

template <int CH, int BS> static void t_fill() {
  char nm[64]; snprintf(nm, 64, "fill_wave CH=%d BS=%d", CH, BS);
  RUN(nm, hipLaunchKernelGGL((fill_wave<CH, BS>), dim3(nvec / (BS * CH)), dim3(BS), 0, 0, out, 7u));
}
template <int CH, int BS> static void t_fill_nt() {
  char nm[64]; snprintf(nm, 64, "fill_wave_nt CH=%d BS=%d", CH, BS);
  RUN(nm, hipLaunchKernelGGL((fill_wave_nt<CH, BS>), dim3(nvec / (BS * CH)), dim3(BS), 0, 0, out, 7u));
}
template <int CH, bool DRAIN> static void t_gs_nt(int g) {
  char nm[64]; snprintf(nm, 64, "fill_wave_gs_nt CH=%d drain=%d grid=%d", CH, (int)DRAIN, g);
  RUN(nm, hipLaunchKernelGGL((fill_wave_gs_nt<CH, DRAIN>), dim3(g), dim3(256), 0, 0, out, nvec / (64 * CH), 7u));
}
template <int CH, int BS, int DEP> static void t_dep() {
  char nm[64]; snprintf(nm, 64, "fill_wave_dep CH=%d BS=%d DEP=%d", CH, BS, DEP);
  RUN(nm, hipLaunchKernelGGL((fill_wave_dep<CH, BS, DEP>), dim3(nvec / (BS * CH)), dim3(BS), 0, 0, in, out));
}
template <int CH, bool DRAIN> static void t_gs(int g) {
  char nm[64]; snprintf(nm, 64, "fill_wave_gs CH=%d drain=%d grid=%d", CH, (int)DRAIN, g);
  RUN(nm, hipLaunchKernelGGL((fill_wave_gs<CH, DRAIN>), dim3(g), dim3(256), 0, 0, out, nvec / (64 * CH), 7u));
}

int main(int argc, char **argv) {
  int log2n = argc > 1 ? atoi(argv[1]) : 28;
  size_t n = (size_t)1 << log2n; nvec = n / 4;
  uint32_t *sink;
  CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(in, 0x5b, n * 4));
  gb = n * 4.0 / 1e6;
  t_fill<1, 256>(); t_fill<2, 256>(); t_fill<4, 256>();
  t_fill_nt<1, 256>(); t_fill_nt<2, 256>(); t_fill_nt<4, 256>(); t_fill_nt<2, 128>(); t_fill_nt<1, 512>();
  for (int g : {256, 512, 1024, 2048}) { t_gs<1, true>(g); t_gs_nt<1, true>(g); t_gs_nt<1, false>(g); t_gs_nt<2, true>(g); t_gs_nt<4, true>(g); }
  return 0;
}
