/*
 * A posting list on the GPU, in plain C: what example.c of lemire/simdcomp does with one block per call
 * (compress() :56-75, the decode loop :101-110, and a search in the compressed blocks), done here with one
 * library call per step over ALL blocks of a device-resident array.
 *
 *   cc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I../include device_postings.c \
 *      -L.. -lsimdcomp_hip -L/opt/rocm/lib -lamdhip64 -o device_postings
 *
 * The byte stream it produces (one width byte + 16*b payload bytes per block of 128) is the one example.c
 * writes: download it and the reference's decode loop reads it, and the other way round.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "simdcomp_hip.h"

#define HIP(call)                                                                              \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s:%d: %s: %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_));    \
      exit(2);                                                                                 \
    }                                                                                          \
  } while (0)
#define SC(call)                                                                               \
  do {                                                                                         \
    int r_ = (call);                                                                           \
    if (r_ != SIMDCOMP_HIP_OK) {                                                               \
      fprintf(stderr, "%s:%d: %s: %d (%s)\n", __FILE__, __LINE__, #call, r_, simdcomp_hip_last_error()); \
      exit(2);                                                                                 \
    }                                                                                          \
  } while (0)

static uint32_t rng_state = 12345;
static uint32_t rng(void) {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 17;
  rng_state ^= rng_state << 5;
  return rng_state;
}

int main(int argc, char **argv) {
  const size_t nblocks = argc > 1 ? strtoull(argv[1], NULL, 10) : 8192;
  const size_t n = nblocks * 128, nqueries = 1000;
  size_t i, q;

  /* sorted document ids: gaps of a few bits, wider in some stretches */
  uint32_t *ids = (uint32_t *)malloc(n * sizeof(uint32_t));
  uint32_t last = 0;
  for (i = 0; i < n; ++i) {
    const unsigned width = 1 + ((i / 128) % 11);
    last += rng() & ((1u << width) - 1);
    ids[i] = last;
  }

  uint32_t *d_ids, *d_out, *d_offsets;
  uint8_t *d_stream, *d_widths;
  uint64_t *d_bytes, *d_result, *d_payload_offsets;
  void *d_ws_compress, *d_ws_decode, *d_ws_index, *d_ws_chain;
  HIP(hipMalloc((void **)&d_ids, n * 4));
  HIP(hipMalloc((void **)&d_out, n * 4));
  HIP(hipMalloc((void **)&d_stream, 513 * nblocks)); /* worst case, as example.c:84 sizes it */
  HIP(hipMalloc((void **)&d_bytes, 8));
  HIP(hipMalloc((void **)&d_result, 3 * 8));
  HIP(hipMalloc(&d_ws_compress, simdcomp_hip_stream_compress_workspace(nblocks) + 16));
  HIP(hipMemcpy(d_ids, ids, n * 4, hipMemcpyHostToDevice));

  /* compress(): widths, offsets and packing of every block, one call */
  SC(simdcomp_hip_stream_compress(SIMDCOMP_HIP_D1, 0, d_ids, nblocks, d_stream, d_bytes, d_ws_compress, NULL));
  uint64_t stream_bytes = 0;
  HIP(hipMemcpy(&stream_bytes, d_bytes, 8, hipMemcpyDeviceToHost));
  printf("%zu integers in %zu blocks -> %llu bytes (%.2f bits per integer)\n", n, nblocks,
         (unsigned long long)stream_bytes, 8.0 * (double)stream_bytes / (double)n);

  /* the decode loop: from the bytes alone, one call */
  HIP(hipMalloc(&d_ws_decode, simdcomp_hip_stream_decode_workspace(stream_bytes, nblocks) + 16));
  SC(simdcomp_hip_stream_decode(SIMDCOMP_HIP_D1, 0, d_stream, stream_bytes, nblocks, d_out, d_result, d_ws_decode, NULL));
  uint64_t result[3];
  uint32_t *back = (uint32_t *)malloc(n * 4);
  HIP(hipMemcpy(result, d_result, sizeof result, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(back, d_out, n * 4, hipMemcpyDeviceToHost));
  if (result[0] != nblocks || result[2] != 0 || memcmp(back, ids, n * 4) != 0) {
    fprintf(stderr, "decode differs from the input\n");
    return 1;
  }
  printf("decoded %llu records, identical to the input\n", (unsigned long long)result[0]);

  /* searching without decoding to memory: index the records once, then batches of (block, key) queries */
  HIP(hipMalloc((void **)&d_widths, nblocks));
  HIP(hipMalloc((void **)&d_payload_offsets, nblocks * 8));
  HIP(hipMalloc((void **)&d_offsets, nblocks * 4));
  HIP(hipMalloc(&d_ws_index, simdcomp_hip_stream_index_workspace(stream_bytes) + 16));
  HIP(hipMalloc(&d_ws_chain, simdcomp_hip_chain_workspace(nblocks) + 16));
  SC(simdcomp_hip_stream_index(d_stream, stream_bytes, nblocks, d_widths, d_payload_offsets, d_result, d_ws_index, NULL));
  SC(simdcomp_hip_chain_offsets(0, d_stream, d_payload_offsets, d_widths, nblocks, 1, d_offsets, d_ws_chain, NULL));

  uint64_t *blocks = (uint64_t *)malloc(nqueries * 8), *d_blocks;
  uint32_t *keys = (uint32_t *)malloc(nqueries * 4), *d_keys, *d_values;
  int32_t *d_positions;
  for (q = 0; q < nqueries; ++q) {
    blocks[q] = rng() % nblocks;
    const uint32_t lo = blocks[q] ? ids[blocks[q] * 128 - 1] : 0, hi = ids[blocks[q] * 128 + 127];
    keys[q] = lo + (hi > lo ? rng() % (hi - lo + 1) : 0);
  }
  HIP(hipMalloc((void **)&d_blocks, nqueries * 8));
  HIP(hipMalloc((void **)&d_keys, nqueries * 4));
  HIP(hipMalloc((void **)&d_values, nqueries * 4));
  HIP(hipMalloc((void **)&d_positions, nqueries * 4));
  HIP(hipMemcpy(d_blocks, blocks, nqueries * 8, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_keys, keys, nqueries * 4, hipMemcpyHostToDevice));
  SC(simdcomp_hip_search(SIMDCOMP_HIP_D1, 0, d_offsets, d_stream, d_payload_offsets, d_widths, 0, 1, d_blocks, d_keys, NULL,
                         nqueries, d_positions, d_values, NULL, NULL));
  int32_t *positions = (int32_t *)malloc(nqueries * 4);
  uint32_t *values = (uint32_t *)malloc(nqueries * 4);
  HIP(hipMemcpy(positions, d_positions, nqueries * 4, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(values, d_values, nqueries * 4, hipMemcpyDeviceToHost));
  for (q = 0; q < nqueries; ++q) {
    /* first value of the block that is >= key (the key never exceeds the block's last value here) */
    const uint32_t *block = ids + blocks[q] * 128;
    int want = 0;
    while (block[want] < keys[q]) ++want;
    if (positions[q] != want || values[q] != block[want]) {
      fprintf(stderr, "query %zu: block %llu key %u: got (%d, %u), want (%d, %u)\n", q, (unsigned long long)blocks[q], keys[q],
              positions[q], values[q], want, block[want]);
      return 1;
    }
  }
  printf("%zu searches in the compressed blocks agree with a scan of the input\n", nqueries);
  printf("ok\n");
  return 0;
}
