// Launchers for the callers either side of the block codec: the byte stream of
// example.c:56-75 ([1 byte b][16*b bytes payload] per block) and the chained
// differential decode of example.c:101-110 that stores no per-block offsets.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace sc {

// --- chained d1 seeds: seed[0] = init, seed[k+1] = last value of block k ------------------------------------------
size_t chain_workspace_bytes(size_t nblocks);
// stream_layout: `payload` is the byte stream (payload of block k at off[k] + k + 1), else the aligned arena
hipError_t launch_chain_offsets(uint32_t init, const void *payload, const uint64_t *off, const uint8_t *bits, size_t nblocks,
                                bool stream_layout, uint32_t *seeds_out, void *workspace, hipStream_t st);

// --- [b][payload] byte stream <-> 16-byte aligned arena -----------------------------------------------------------
// record k of the stream starts at byte off[k] + k (off[] = exclusive scan of 16*bits[])
hipError_t launch_stream_build(const void *arena, const uint64_t *off, const uint8_t *bits, size_t nblocks, uint8_t *stream,
                               hipStream_t st);
hipError_t launch_stream_unbuild(const uint8_t *stream, const uint64_t *off, const uint8_t *bits, size_t nblocks, void *arena,
                                 hipStream_t st);

// *out = *payload_bytes + nblocks: the length of the stream that holds nblocks records
hipError_t launch_stream_bytes(const uint64_t *payload_bytes, size_t nblocks, uint64_t *out, hipStream_t st);

// --- finding the records of a stream whose widths are only known from the stream itself ---------------------------
size_t stream_index_workspace_bytes(size_t stream_bytes);
// result[0] = records found, result[1] = arena bytes (16 * sum of widths), result[2] = 0 when the stream parsed
// cleanly to its last byte, 1 otherwise (a width byte above 32 or a truncated last record).  At most max_blocks
// entries of bits[] / off[] are written; with clear_tail the entries among those max_blocks that belong to no record
// (a short stream; a bad one beyond its good part) are set to zero as well, otherwise they are left alone.
hipError_t launch_stream_index(const uint8_t *stream, size_t stream_bytes, size_t max_blocks, uint8_t *bits, uint64_t *off,
                               uint64_t *result, void *workspace, bool clear_tail, hipStream_t st);

}  // namespace sc
