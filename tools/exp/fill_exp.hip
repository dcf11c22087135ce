// Development harness: which store pattern reaches the HBM write ceiling?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// grid-stride, thread-interleaved
__global__ __launch_bounds__(256) void fill_gs(u32x4 *out, size_t nvec, uint32_t val) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) out[i] = u32x4{val, val, val, val};
}
// one shot: each workgroup owns CH vectors per thread, interleaved inside the chunk
template <int CH, int BS>
__global__ __launch_bounds__(BS) void fill_chunk(u32x4 *out, size_t nvec, uint32_t val) {
  size_t base = (size_t)blockIdx.x * BS * CH + threadIdx.x;
#pragma unroll
  for (int j = 0; j < CH; ++j) out[base + (size_t)j * BS] = u32x4{val, val, val, val};
}
// one shot: each thread owns CH consecutive vectors
template <int CH, int BS>
__global__ __launch_bounds__(BS) void fill_thread(u32x4 *out, size_t nvec, uint32_t val) {
  size_t base = ((size_t)blockIdx.x * BS + threadIdx.x) * CH;
#pragma unroll
  for (int j = 0; j < CH; ++j) out[base + j] = u32x4{val, val, val, val};
}
// persistent, chunked: workgroup loops over chunks of BS*CH vectors
template <int CH, int BS>
__global__ __launch_bounds__(BS) void fill_chunk_gs(u32x4 *out, size_t nvec, uint32_t val) {
  const size_t nch = nvec / (BS * CH);
  for (size_t c = blockIdx.x; c < nch; c += gridDim.x) {
    size_t base = c * BS * CH + threadIdx.x;
#pragma unroll
    for (int j = 0; j < CH; ++j) out[base + (size_t)j * BS] = u32x4{val, val, val, val};
  }
}
template <int CH, int BS>
__global__ __launch_bounds__(BS) void copy_chunk(const u32x4 *in, u32x4 *out, size_t nvec) {
  size_t base = (size_t)blockIdx.x * BS * CH + threadIdx.x;
  u32x4 v[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) v[j] = in[base + (size_t)j * BS];
#pragma unroll
  for (int j = 0; j < CH; ++j) out[base + (size_t)j * BS] = v[j];
}
template <int CH, int BS>
__global__ __launch_bounds__(BS) void read_chunk(const u32x4 *in, uint32_t *sink, size_t nvec) {
  size_t base = (size_t)blockIdx.x * BS * CH + threadIdx.x;
  u32x4 v[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) v[j] = in[base + (size_t)j * BS];
  uint32_t acc = 0;
#pragma unroll
  for (int j = 0; j < CH; ++j) acc |= v[j].x | v[j].y | v[j].z | v[j].w;
  if (acc == 0x12345678u) sink[0] = acc;
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

int main(int argc, char **argv) {
  int log2n = argc > 1 ? atoi(argv[1]) : 28;
  size_t n = (size_t)1 << log2n, nvec = n / 4;
  u32x4 *out, *in; uint32_t *sink;
  CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(in, 0x5b, n * 4));
  double gb = n * 4.0 / 1e6;
#define RUN(name, ...) { double ms = time_ms([&] { __VA_ARGS__; }); printf("%-34s %.4f ms %8.1f GB/s\n", name, ms, gb / ms); }
  for (int g : {1024, 2048, 4096, 8192}) { char nm[64]; snprintf(nm, 64, "fill_gs grid=%d", g); RUN(nm, hipLaunchKernelGGL(fill_gs, dim3(g), dim3(256), 0, 0, out, nvec, 7u)); }
  RUN("fill_chunk<1,256>", hipLaunchKernelGGL((fill_chunk<1, 256>), dim3(nvec / 256), dim3(256), 0, 0, out, nvec, 7u));
  RUN("fill_chunk<4,256>", hipLaunchKernelGGL((fill_chunk<4, 256>), dim3(nvec / 1024), dim3(256), 0, 0, out, nvec, 7u));
  RUN("fill_chunk<8,256>", hipLaunchKernelGGL((fill_chunk<8, 256>), dim3(nvec / 2048), dim3(256), 0, 0, out, nvec, 7u));
  RUN("fill_chunk<16,256>", hipLaunchKernelGGL((fill_chunk<16, 256>), dim3(nvec / 4096), dim3(256), 0, 0, out, nvec, 7u));
  RUN("fill_chunk<4,512>", hipLaunchKernelGGL((fill_chunk<4, 512>), dim3(nvec / 2048), dim3(512), 0, 0, out, nvec, 7u));
  RUN("fill_chunk<4,1024>", hipLaunchKernelGGL((fill_chunk<4, 1024>), dim3(nvec / 4096), dim3(1024), 0, 0, out, nvec, 7u));
  RUN("fill_chunk<4,64>", hipLaunchKernelGGL((fill_chunk<4, 64>), dim3(nvec / 256), dim3(64), 0, 0, out, nvec, 7u));
  RUN("fill_thread<4,256>", hipLaunchKernelGGL((fill_thread<4, 256>), dim3(nvec / 1024), dim3(256), 0, 0, out, nvec, 7u));
  for (int g : {1024, 2048, 4096, 8192}) { char nm[64]; snprintf(nm, 64, "fill_chunk_gs<4,256> grid=%d", g); RUN(nm, hipLaunchKernelGGL((fill_chunk_gs<4, 256>), dim3(g), dim3(256), 0, 0, out, nvec, 7u)); }
  for (int g : {1024, 2048, 4096}) { char nm[64]; snprintf(nm, 64, "fill_chunk_gs<16,256> grid=%d", g); RUN(nm, hipLaunchKernelGGL((fill_chunk_gs<16, 256>), dim3(g), dim3(256), 0, 0, out, nvec, 7u)); }
  gb *= 2;
  RUN("copy_chunk<1,256>", hipLaunchKernelGGL((copy_chunk<1, 256>), dim3(nvec / 256), dim3(256), 0, 0, in, out, nvec));
  RUN("copy_chunk<4,256>", hipLaunchKernelGGL((copy_chunk<4, 256>), dim3(nvec / 1024), dim3(256), 0, 0, in, out, nvec));
  RUN("copy_chunk<8,256>", hipLaunchKernelGGL((copy_chunk<8, 256>), dim3(nvec / 2048), dim3(256), 0, 0, in, out, nvec));
  RUN("copy_chunk<4,1024>", hipLaunchKernelGGL((copy_chunk<4, 1024>), dim3(nvec / 4096), dim3(1024), 0, 0, in, out, nvec));
  { double ms = time_ms([&] { CK(hipMemcpyAsync(out, in, n * 4, hipMemcpyDeviceToDevice, 0)); }); printf("%-34s %.4f ms %8.1f GB/s\n", "hipMemcpy D2D", ms, gb / ms); }
  gb /= 2;
  RUN("read_chunk<4,256>", hipLaunchKernelGGL((read_chunk<4, 256>), dim3(nvec / 1024), dim3(256), 0, 0, in, sink, nvec));
  RUN("read_chunk<8,256>", hipLaunchKernelGGL((read_chunk<8, 256>), dim3(nvec / 2048), dim3(256), 0, 0, in, sink, nvec));
  { double ms = time_ms([&] { CK(hipMemsetAsync(out, 0, n * 4, 0)); }); printf("%-34s %.4f ms %8.1f GB/s\n", "hipMemset", ms, gb / ms); }
  return 0;
}
