// Internal interface between the kernel translation unit and the C-ABI layers.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace sc {

enum { MODE_PLAIN = 0, MODE_D1 = 1, MODE_FOR = 2 };
enum { RED_MAXBITS = 0, RED_MAXBITSD1 = 1, RED_MINMAX = 2 };

struct UnpackArgs {
  const void *payload;     // 16-byte aligned
  const uint64_t *off;     // per-block payload byte offsets (multiples of 16) or null: block k at k*16*bit
  const uint8_t *bits;     // per-block widths or null: uniform `bit`
  const uint32_t *seeds;   // per-block d1 offset / FOR base or null: `seed0` for every block
  uint32_t *out;
  size_t nblocks;
  size_t length;           // TAIL kernels: number of valid integers
  uint32_t bit;
  uint32_t seed0;
  // Per-block widths only: the payload of block k lies at payload + off[k] + (block_base + k) + 1, any alignment --
  // the byte stream of example.c:56-75 (a width byte in front of every payload) decoded in place.
  uint32_t stream_layout;
  size_t block_base;       // stream layout: record number of this call's block 0
  // Wide layouts (uniform width, plain mode, flex kernels only): an AVX2 / AVX-512 block (avxbitpacking.c:9906,
  // avx512bitpacking.c) is lane_groups = 2 / 4 interleaved 4-lane blocks -- group g holds lanes 4g..4g+3, its row k
  // is the 4 integers at 4*lane_groups*k + 4g of the wide block and its vector w the 16 bytes at
  // 16*lane_groups*w + 16g of the wide payload.  nblocks counts the 4-lane blocks.  0 or 1: the plain layout.
  uint32_t lane_groups;
};

struct PackArgs {
  const uint32_t *in;
  void *payload;
  const uint64_t *off;
  const uint8_t *bits;
  const uint32_t *seeds;   // d1: null = chain from the input (block k seeded by in[128k-1], block 0 by seed0)
  uint32_t *seeds_out;     // d1: optional, receives the seed used for every block
  size_t nblocks;
  size_t length;           // GENERIC kernels: number of valid integers
  uint32_t bit;
  uint32_t seed0;
  int carry_blocks;        // GENERIC: hand the end-of-block spill to the next block (whole-array generic loop)
  size_t chain_base;       // d1 chain mode: index of this call's block 0 in the whole stream (0 = stream start)
  int carry_first;         // GENERIC: block 0 also receives a spill, from the row stored just before `in`
  // Per-block widths only: write the byte stream of example.c:56-75 directly -- block k's width byte at
  // payload + off[k] + (chain_base + k), its payload right behind it, any alignment.
  uint32_t stream_layout;
  uint32_t lane_groups;    // as in UnpackArgs
};

struct ReduceArgs {
  uint32_t group;          // RED_MAXBITS: a block is `group` x 128 consecutive integers (0 or 1: 128)
  const uint32_t *in;
  const uint32_t *seeds;
  uint32_t *seeds_out;
  uint8_t *bits_out;
  uint32_t *min_out;
  uint32_t *max_out;
  size_t nblocks;
  uint32_t seed0;
};

hipError_t launch_unpack(int mode, bool tail, const UnpackArgs &a, hipStream_t st);
hipError_t launch_pack(int mode, bool masked, bool generic, const PackArgs &a, hipStream_t st);
hipError_t launch_reduce(int what, const ReduceArgs &a, hipStream_t st);
size_t scan_workspace_bytes(size_t nblocks);
hipError_t launch_payload_offsets(const uint8_t *bits, size_t nblocks, uint64_t *off, uint64_t *total,
                                  void *workspace, hipStream_t st);
hipError_t launch_reduce_array(const uint32_t *in, size_t length, uint32_t init, uint32_t *result4, hipStream_t st);
hipError_t launch_checksum(const uint32_t *data, const uint32_t *other, size_t nwords, uint64_t index0,
                           uint64_t *acc, hipStream_t st);

}  // namespace sc
