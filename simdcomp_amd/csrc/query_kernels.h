// Batched random access into packed blocks: many (block, slot) or (block, key)
// queries per launch -- the device form of simdselectd1 / simdsearchwithlengthd1 /
// simdsearchd1 / simdscand1 (src/simdpackedselect.c, src/simdpackedsearch.c) and
// simdselectFOR / simdsearchwithlengthFOR (src/simdfor.c:14987, :15016).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace sc {

enum { Q_SELECT = 0, Q_SEARCH = 1, Q_SCAN = 2 };

struct QueryArgs {
  const void *payload;
  const uint64_t *off;           // per-block payload offsets or null: block k at k*16*bit
  const uint8_t *bits;           // per-block widths or null: uniform `bit`
  const uint32_t *seeds;         // per-block d1 offset / FOR base or null: seed0
  uint32_t bit;
  uint32_t seed0;
  uint32_t stream_layout;        // with off: payload of block k at off[k] + k + 1 (the example.c byte stream)
  const uint64_t *query_blocks;  // block of query q, or null: query q addresses block q
  const int32_t *slots;          // Q_SELECT
  const uint32_t *keys;          // Q_SEARCH
  const int32_t *lengths;        // Q_SEARCH: values considered per block, or null: 128
  size_t nqueries;
  uint32_t *values_out;
  int32_t *positions_out;
  uint32_t *last4_out;           // d1: 4 words per query, the offset vector simdsearchd1 / simdscand1 leave behind
  // instead of off / bits / seeds: one 16-byte entry per block (launch_block_directory writes them), so that a
  // query on a block nobody else asks for costs one scattered metadata read, not three
  const void *directory;
};

// directory entry of block k: word 0 = low half of the payload's byte offset from `payload` (the stream layout's
// k + 1 width bytes included), word 1 = bits 32..47 of that offset | width << 24, word 2 = seed, word 3 = 0
constexpr size_t DIRECTORY_ENTRY_BYTES = 16;
constexpr uint64_t DIRECTORY_MAX_OFFSET = (1ull << 48) - 1;

hipError_t launch_query(int mode, int what, const QueryArgs &a, hipStream_t st);
hipError_t launch_block_directory(uint32_t seed0, const uint32_t *seeds, const uint64_t *off, const uint8_t *bits, size_t nblocks,
                                  uint32_t stream_layout, void *directory, hipStream_t st);

}  // namespace sc
