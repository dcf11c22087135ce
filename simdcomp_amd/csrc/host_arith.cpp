// Entry points of include/simdcomp_hip.h that are host arithmetic only -- sizes, block-range shards, the version
// and the error text: no HIP call, no device.  Kept in a translation unit of their own so that they also build
// with the host compiler alone (the sanitizer build, `make asan`).
#include <stdio.h>

#include "../../include/simdcomp_hip.h"
#include "api_error.h"

namespace {
thread_local char g_err[256] = "";
int fail(int code, const char *what) { return sc::api_fail(code, what); }
}  // namespace

namespace sc {
int api_fail(int code, const char *what) {
  snprintf(g_err, sizeof g_err, "%s", what);
  return code;
}
}  // namespace sc

extern "C" {

int simdcomp_hip_version(void) { return 100; }

const char *simdcomp_hip_last_error(void) { return g_err; }

// src/simdbitpacking.c:14370, src/simdfor.c:15034 in size_t arithmetic
size_t simdcomp_hip_compressedbytes(size_t length, uint32_t bit) {
  if (bit == 0) return 0;
  if (bit >= 32) return length * 4;
  return (((length + 3) / 4) * (size_t)bit + 31) / 32 * 16;
}

// Blocks are independent (src/simdbitpacking.c:14382-14386 carries no state from one block to the next): a corpus
// is sharded by contiguous block ranges.
int simdcomp_hip_partition(size_t nblocks, int nparts, int part, size_t *first, size_t *count) {
  if (nparts <= 0 || part < 0 || part >= nparts || !first || !count) return fail(SIMDCOMP_HIP_ESIZE, "bad partition");
  const size_t lo = (size_t)((unsigned __int128)nblocks * (unsigned)part / (unsigned)nparts);
  const size_t hi = (size_t)((unsigned __int128)nblocks * (unsigned)(part + 1) / (unsigned)nparts);
  *first = lo;
  *count = hi - lo;
  return SIMDCOMP_HIP_OK;
}

int simdcomp_hip_partition_by_bytes(const uint64_t *h_payload_offsets, size_t nblocks, int nparts, int part,
                                    size_t *first, size_t *count) {
  if (nparts <= 0 || part < 0 || part >= nparts || !first || !count || (!h_payload_offsets && nblocks))
    return fail(SIMDCOMP_HIP_ESIZE, "bad partition");
  // cost of blocks [0, k): payload bytes read + 512 bytes written per block
  auto cost = [&](size_t k) -> unsigned __int128 {
    return (unsigned __int128)(h_payload_offsets[k] - h_payload_offsets[0]) + (unsigned __int128)k * 512u;
  };
  auto cut = [&](int p) -> size_t {
    if (p <= 0) return 0;
    if (p >= nparts) return nblocks;
    const unsigned __int128 target = cost(nblocks) * (unsigned)p / (unsigned)nparts;
    size_t lo = 0, hi = nblocks;  // first k with cost(k) >= target
    while (lo < hi) {
      const size_t mid = lo + (hi - lo) / 2;
      if (cost(mid) < target) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  const size_t a = nblocks ? cut(part) : 0, b = nblocks ? cut(part + 1) : 0;
  *first = a;
  *count = b - a;
  return SIMDCOMP_HIP_OK;
}

}  // extern "C"
