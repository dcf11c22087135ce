/*
 * simdcomp_hip.h -- batched, device-pointer C ABI of the MI355X block codec.
 *
 * Every entry point is the "many blocks per call" form of one routine of
 * lemire/simdcomp's C API; the comment on each names the reference routine
 * (file:line under the reference tree) whose per-block behaviour it reproduces
 * bit for bit.  The compressed layout is the reference's own: b 16-byte vectors
 * per 128-integer block, four interleaved lanes (include/simdcomputil.h:22,
 * src/simdbitpacking.c:217-323).
 *
 * Conventions
 *   - all d_* pointers are DEVICE pointers on the current HIP device;
 *     integer arrays and payloads must be 16-byte aligned, per-block payload
 *     offsets are byte offsets and must be multiples of 16;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls
 *     only enqueue work, they never synchronise, allocate or free;
 *   - sizes are size_t (the reference's int lengths overflow at 2^30 values);
 *   - return value: 0 on success, a positive hipError_t if a HIP call failed,
 *     or one of the negative SIMDCOMP_HIP_E* codes for a rejected argument.
 *     A call that launches resets the calling thread's HIP error state first
 *     (hipGetLastError), so that an earlier failure of the application's own is
 *     not reported as this call's.
 *     The reference has no error channel (bit > 32 silently does nothing,
 *     src/simdbitpacking.c:14046); here bit > 32 is SIMDCOMP_HIP_EBIT.
 *   - a block's "initvalue" is the reference's initvalue argument: the
 *     differential seed for the d1 routines, the frame base for the FOR ones;
 *   - per-block widths (d_bits[k]) above 32 are read as 32 by every kernel.
 */
#ifndef SIMDCOMP_HIP_H_
#define SIMDCOMP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIMDCOMP_HIP_OK 0
#define SIMDCOMP_HIP_EBIT (-1)   /* bit width > 32 */
#define SIMDCOMP_HIP_EALIGN (-2) /* pointer or offset not 16-byte aligned */
#define SIMDCOMP_HIP_ENULL (-3)  /* required pointer is NULL */
#define SIMDCOMP_HIP_EMODE (-4)  /* unknown mode */
#define SIMDCOMP_HIP_ESIZE (-5)  /* size beyond what the entry point supports */

#define SIMDCOMP_HIP_PLAIN 0 /* simdpack / simdunpack family */
#define SIMDCOMP_HIP_D1 1    /* simdpackd1 / simdunpackd1 family */
#define SIMDCOMP_HIP_FOR 2   /* simdpackFOR / simdunpackFOR family */

/* Library version (major*10000 + minor*100 + patch). */
int simdcomp_hip_version(void);
/* Text of the most recent failure on the calling thread ("" if none). */
const char *simdcomp_hip_last_error(void);

/* The reference-signature host routines of simdcomp.h (simdpack, simdunpackd1,
 * simdpack_length, ...) return void or a pointer, as in the reference
 * (include/simdbitpacking.h:24-68): they cannot report that the GPU path failed
 * (no device, out of memory, a HIP error).  By default such a failure prints to
 * stderr and aborts.  A program may install a handler instead; it receives the
 * message and `arg` and must not return (longjmp, exit, or throw through C++
 * frames); if it does return the library aborts.  NULL restores the default.
 * Before the handler runs the device is drained, so no copy of the failed call
 * is still writing into the caller's buffers (their contents are unspecified).
 * After a handler has left through longjmp the calling thread's current HIP
 * device may be the one the call ran on (SIMDCOMP_HIP_DEVICES) rather than the
 * caller's; set it again before other HIP work.  The library stays usable.
 * The batched entry points below never call it: they return error codes. */
typedef void (*simdcomp_hip_error_handler)(const char *message, void *arg);
void simdcomp_hip_set_error_handler(simdcomp_hip_error_handler handler, void *arg);

/* ---- uniform width: block k's payload starts at d_out + k*16*bit ---------- */

/* nblocks x simdpack (masked != 0, src/simdbitpacking.c:14052) or
 * simdpackwithoutmask (masked == 0, src/simdbitpacking.c:13913). */
int simdcomp_hip_pack(const uint32_t *d_in, void *d_out, size_t nblocks, uint32_t bit, int masked,
                      void *stream);
/* nblocks x simdunpack (src/simdbitpacking.c:13773). */
int simdcomp_hip_unpack(const void *d_in, uint32_t *d_out, size_t nblocks, uint32_t bit, void *stream);

/* nblocks x simdpackd1 (masked, src/simdintegratedbitpacking.c:25198) or
 * simdpackwithoutmaskd1 (:25058).  d_initvalues[k] seeds block k.  If
 * d_initvalues is NULL the blocks are chained the way example.c:63-71 does it:
 * block 0 starts from `initvalue`, block k from d_in[128k-1]; `initvalue` is
 * ignored otherwise.  d_initvalues_out (optional) receives the seed used for
 * each block, i.e. the array simdcomp_hip_unpackd1 needs. */
int simdcomp_hip_packd1(uint32_t initvalue, const uint32_t *d_initvalues, const uint32_t *d_in, void *d_out,
                        size_t nblocks, uint32_t bit, int masked, uint32_t *d_initvalues_out, void *stream);
/* nblocks x simdunpackd1 (src/simdintegratedbitpacking.c:24915); block k is
 * decoded from seed d_initvalues[k], no block depends on another. */
int simdcomp_hip_unpackd1(const uint32_t *d_initvalues, const void *d_in, uint32_t *d_out, size_t nblocks,
                          uint32_t bit, void *stream);

/* nblocks x simdpackFOR (src/simdfor.c:14705), frame base d_initvalues[k]. */
int simdcomp_hip_packFOR(const uint32_t *d_initvalues, const uint32_t *d_in, void *d_out, size_t nblocks,
                         uint32_t bit, void *stream);
/* nblocks x simdunpackFOR (src/simdfor.c:14846). */
int simdcomp_hip_unpackFOR(const uint32_t *d_initvalues, const void *d_in, uint32_t *d_out, size_t nblocks,
                           uint32_t bit, void *stream);

/* ---- per-block widths: the caller convention of example.c:56-75,135-150 ----
 * d_bits[k] is block k's width, d_payload_offsets[k] the byte offset of its
 * payload inside d_payload (see simdcomp_hip_payload_offsets; d_payload is ONE
 * buffer: the decoder may also read bytes that lie between two payloads, ahead
 * of decoding them).  mode is one of
 * SIMDCOMP_HIP_PLAIN / _D1 / _FOR; masked and the initvalue arguments are as in
 * the uniform calls (FOR ignores masked). */
int simdcomp_hip_pack_mixed(int mode, int masked, uint32_t initvalue, const uint32_t *d_initvalues,
                            const uint32_t *d_in, void *d_payload, const uint64_t *d_payload_offsets,
                            const uint8_t *d_bits, size_t nblocks, uint32_t *d_initvalues_out, void *stream);
int simdcomp_hip_unpack_mixed(int mode, const uint32_t *d_initvalues, const void *d_payload,
                              const uint64_t *d_payload_offsets, const uint8_t *d_bits, uint32_t *d_out,
                              size_t nblocks, void *stream);

/* ---- per-block reductions -------------------------------------------------- */

/* d_bits[k] = maxbits(block k) (src/simdcomputil.c:42). */
int simdcomp_hip_maxbits(const uint32_t *d_in, size_t nblocks, uint8_t *d_bits, void *stream);
/* d_bits[k] = simdmaxbitsd1(seed k, block k) (src/simdcomputil.c:181); seeds as
 * in simdcomp_hip_packd1 (explicit array, or chained from the input). */
int simdcomp_hip_maxbitsd1(uint32_t initvalue, const uint32_t *d_initvalues, const uint32_t *d_in,
                           size_t nblocks, uint8_t *d_bits, uint32_t *d_initvalues_out, void *stream);
/* simdmaxmin per block (src/simdcomputil.c:95): d_min[k], d_max[k] (either may
 * be NULL) and, if d_bits is given, d_bits[k] = bits(max - min), the frame
 * width a FOR caller picks (tests/unit.c:659-666). */
int simdcomp_hip_maxmin(const uint32_t *d_in, size_t nblocks, uint32_t *d_min, uint32_t *d_max,
                        uint8_t *d_bits, void *stream);

/* Whole-array reductions behind maxbits_length (src/simdcomputil.c:160),
 * simdmaxbitsd1_length (:198), simdmin_length (:109) and simdmaxmin_length
 * (:130): d_result4[0] = OR of the values, [1] = OR of successive differences
 * seeded by initvalue, [2] = minimum, [3] = maximum (an empty array gives
 * 0, 0, 0xFFFFFFFF, 0).  bits() of [0] / [1] are the two widths.  d_in needs
 * only 4-byte alignment. */
int simdcomp_hip_reduce_length(const uint32_t *d_in, size_t length, uint32_t initvalue, uint32_t *d_result4,
                               void *stream);

/* Exclusive prefix sum of 16*d_bits[k] -> d_payload_offsets[k]; *d_total_bytes
 * (optional) receives the payload size.  d_workspace must hold
 * simdcomp_hip_payload_offsets_workspace(nblocks) bytes. */
size_t simdcomp_hip_payload_offsets_workspace(size_t nblocks);
int simdcomp_hip_payload_offsets(const uint8_t *d_bits, size_t nblocks, uint64_t *d_payload_offsets,
                                 uint64_t *d_total_bytes, void *d_workspace, void *stream);

/* simdcomp_hip_payload_offsets for callers that cannot allocate device memory or
 * carry a workspace (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference, which never
 * allocates (it has no scan: its callers advance a pointer, example.c:72).  Same
 * kernels and same results as simdcomp_hip_payload_offsets; the scan's tile sums
 * go to a 1 MiB device array that belongs to the library (one copy per GPU, held
 * until the process ends), so the call performs no hipMalloc / hipFree.  Its
 * device address is resolved on the first call on each device (make that call
 * outside graph capture).  Limit: nblocks <=
 * simdcomp_hip_payload_offsets_noalloc_max_blocks() (2^29 blocks = 2^36
 * integers), beyond that SIMDCOMP_HIP_ESIZE.  Calls on the same GPU share the
 * array: two of them in flight at once (different streams or threads) race. */
size_t simdcomp_hip_payload_offsets_noalloc_max_blocks(void);
int simdcomp_hip_payload_offsets_noalloc(const uint8_t *d_bits, size_t nblocks, uint64_t *d_payload_offsets,
                                         uint64_t *d_total_bytes, void *stream);

/* ---- arbitrary lengths (device-resident forms of the *_length routines) ---- */

/* simdpack_compressedbytes / simdpackFOR_compressedbytes
 * (src/simdbitpacking.c:14370, src/simdfor.c:15034) in size_t arithmetic. */
size_t simdcomp_hip_compressedbytes(size_t length, uint32_t bit);
/* simdpack_length (src/simdbitpacking.c:14379): whole blocks masked, the
 * remainder through the unmasked short packer; writes
 * simdcomp_hip_compressedbytes(length, bit) bytes. */
int simdcomp_hip_pack_length(const uint32_t *d_in, size_t length, void *d_out, uint32_t bit, void *stream);
/* simdunpack_length (src/simdbitpacking.c:14390). */
int simdcomp_hip_unpack_length(const void *d_in, size_t length, uint32_t *d_out, uint32_t bit, void *stream);
/* simdpackFOR_length (src/simdfor.c:15043): one frame base for the whole array.
 * carry_in != 0 says that d_in[-4..-1] hold the row that precedes this piece of
 * a longer array (used when an array is packed piecewise; 0 for a whole array). */
int simdcomp_hip_packFOR_length(uint32_t initvalue, const uint32_t *d_in, size_t length, void *d_out,
                                uint32_t bit, int carry_in, void *stream);
/* simdunpackFOR_length (src/simdfor.c:15098). */
int simdcomp_hip_unpackFOR_length(uint32_t initvalue, const void *d_in, size_t length, uint32_t *d_out,
                                  uint32_t bit, void *stream);

/* ---- the callers either side of the block codec ----------------------------- */

/* Chained differential decode WITHOUT stored per-block offsets, the decode loop
 * of example.c:101-110 (`offset = backbuffer[SIMDBlockSize - 1]` after every
 * block): computes d_initvalues_out[k] = the offset simdunpackd1 must be given
 * for block k when block 0 starts from `initvalue`.  A 32-bit block holds raw
 * values and restarts the chain at its last one, a 0-bit block passes the
 * offset on (src/simdintegratedbitpacking.c:21-35, :24903-24913).  Follow with
 * simdcomp_hip_unpack_mixed / _unpack_stream (SIMDCOMP_HIP_D1, d_initvalues_out).
 * stream_layout = 0: d_payload is the aligned arena; 1: it is the byte stream
 * described below.  d_workspace: simdcomp_hip_chain_workspace(nblocks) bytes,
 * 16-byte aligned. */
size_t simdcomp_hip_chain_workspace(size_t nblocks);
int simdcomp_hip_chain_offsets(uint32_t initvalue, const void *d_payload, const uint64_t *d_payload_offsets,
                               const uint8_t *d_bits, size_t nblocks, int stream_layout, uint32_t *d_initvalues_out,
                               void *d_workspace, void *stream);

/* The byte stream example.c:56-75 and :135-150 write: per block one width byte
 * followed by its 16*b payload bytes (`*buffer++ = b; ... buffer += b *
 * sizeof(__m128i)`).  Record k starts at byte d_payload_offsets[k] + k, where
 * d_payload_offsets is the same exclusive scan of 16*b the aligned arena uses;
 * the whole stream is nblocks + (total payload bytes) long; no alignment is
 * required of d_stream (the reference reads and writes it with loadu/storeu,
 * example.c:68-72; here it is byte-addressed 16-byte global accesses).
 *
 * pack_stream / unpack_stream are pack_mixed / unpack_mixed working on such a
 * stream in place (pack_stream also writes the width bytes): with
 * simdcomp_hip_maxbitsd1 and simdcomp_hip_payload_offsets in front,
 * pack_stream is example.c:56-75 (SIMDCOMP_HIP_D1, masked = 0, chained) or
 * :135-150 (SIMDCOMP_HIP_PLAIN after simdcomp_hip_maxbits). */
int simdcomp_hip_pack_stream(int mode, int masked, uint32_t initvalue, const uint32_t *d_initvalues,
                             const uint32_t *d_in, void *d_stream, const uint64_t *d_payload_offsets,
                             const uint8_t *d_bits, size_t nblocks, uint32_t *d_initvalues_out, void *stream);
int simdcomp_hip_unpack_stream(int mode, const uint32_t *d_initvalues, const void *d_stream,
                               const uint64_t *d_payload_offsets, const uint8_t *d_bits, uint32_t *d_out,
                               size_t nblocks, void *stream);
/* build copies an aligned payload arena into such a stream, unbuild copies the
 * payloads of a stream back into an aligned arena. */
int simdcomp_hip_stream_build(const void *d_payload, const uint64_t *d_payload_offsets, const uint8_t *d_bits,
                              size_t nblocks, void *d_stream, void *stream);
int simdcomp_hip_stream_unbuild(const void *d_stream, const uint64_t *d_payload_offsets, const uint8_t *d_bits,
                                size_t nblocks, void *d_payload, void *stream);
/* Finds the records of such a stream from its bytes alone (what the decode
 * loop of example.c:103-110 does one record at a time): writes the width and
 * the arena offset of every record (at most max_blocks of them) and
 * d_result[0] = number of records, d_result[1] = total payload bytes,
 * d_result[2] = 0 if the stream parsed cleanly to its last byte, 1 if it holds
 * a width byte above 32 or ends inside a record (then only d_result[2] is
 * meaningful).  d_workspace: simdcomp_hip_stream_index_workspace(stream_bytes)
 * bytes, 16-byte aligned. */
size_t simdcomp_hip_stream_index_workspace(size_t stream_bytes);
int simdcomp_hip_stream_index(const void *d_stream, size_t stream_bytes, size_t max_blocks, uint8_t *d_bits,
                              uint64_t *d_payload_offsets, uint64_t *d_result, void *d_workspace, void *stream);

/* compress() of example.c:56-75 (SIMDCOMP_HIP_D1: width from simdmaxbitsd1,
 * simdpackwithoutmaskd1, offset = last value of the previous block, first
 * offset `initvalue`) or varying_bit_width_compress() of :135-150
 * (SIMDCOMP_HIP_PLAIN: maxbits, simdpackwithoutmask) as one call: widths,
 * offsets scan and in-place pack enqueued back to back, no host
 * synchronisation.  d_stream needs room for the worst case, 513 * nblocks
 * bytes (what example.c:84 allocates); *d_stream_bytes (device memory)
 * receives the length actually written.  d_workspace:
 * simdcomp_hip_stream_compress_workspace(nblocks) bytes, 16-byte aligned. */
size_t simdcomp_hip_stream_compress_workspace(size_t nblocks);
int simdcomp_hip_stream_compress(int mode, uint32_t initvalue, const uint32_t *d_in, size_t nblocks, void *d_stream,
                                 uint64_t *d_stream_bytes, void *d_workspace, void *stream);
/* The decode loops of example.c:101-110 (SIMDCOMP_HIP_D1) and :170-175
 * (SIMDCOMP_HIP_PLAIN) as one call: record index, chained offsets, in-place
 * unpack.  d_result as for simdcomp_hip_stream_index; when the stream holds
 * fewer than nblocks records the missing blocks decode as zero-width blocks
 * (check d_result[0] == nblocks and d_result[2] == 0).  When d_result[2] is 1
 * (bad width byte, or the stream ends inside a record) d_out is unspecified;
 * nothing outside d_stream[0 .. stream_bytes) is read in any case: a record
 * that does not fit is never indexed.  d_workspace:
 * simdcomp_hip_stream_decode_workspace(stream_bytes, nblocks) bytes. */
size_t simdcomp_hip_stream_decode_workspace(size_t stream_bytes, size_t nblocks);
int simdcomp_hip_stream_decode(int mode, uint32_t initvalue, const void *d_stream, size_t stream_bytes, size_t nblocks,
                               uint32_t *d_out, uint64_t *d_result, void *d_workspace, void *stream);

/* ---- batched random access: many queries per launch -------------------------- */

/* Blocks are addressed like everywhere else: uniform width (d_bits = NULL, block k
 * at k*16*bit) or per-block widths with d_payload_offsets (stream_layout = 1: the
 * byte stream above); d_initvalues = per-block d1 offset / FOR base (NULL: the
 * scalar `initvalue` for every block).  Query q addresses block d_query_blocks[q]
 * (NULL: block q).  mode: SIMDCOMP_HIP_D1, SIMDCOMP_HIP_FOR, or
 * SIMDCOMP_HIP_PLAIN (= FOR with base 0).
 *
 * select: d_values_out[q] = value number d_slots[q] of its block --
 *   simdselectd1 (src/simdpackedselect.c:8096, slot taken mod 128) /
 *   simdselectFOR (src/simdfor.c:14987). */
int simdcomp_hip_select(int mode, uint32_t initvalue, const uint32_t *d_initvalues, const void *d_payload,
                        const uint64_t *d_payload_offsets, const uint8_t *d_bits, uint32_t bit, int stream_layout,
                        const uint64_t *d_query_blocks, const int32_t *d_slots, size_t nqueries,
                        uint32_t *d_values_out, void *stream);
/* search: position and value of the first of the first d_lengths[q] (NULL: 128)
 * values of the block that is >= d_keys[q] --
 *   D1: simdsearchwithlengthd1 (src/simdpackedsearch.c:8284): not found gives
 *       (length, key + 1); a 32-bit block is bisected as the reference does.
 *       d_last4_out (optional, 4 words per query): the offset vector simdsearchd1
 *       (:16583) leaves behind when called with all four elements = the offset.
 *   FOR / PLAIN: simdsearchwithlengthFOR (src/simdfor.c:15016): lower-bound
 *       bisection; the value at the returned slot is reported even when it is
 *       below the key (0 when the slot is 128, where the reference reads past
 *       the block).  d_last4_out must be NULL. */
int simdcomp_hip_search(int mode, uint32_t initvalue, const uint32_t *d_initvalues, const void *d_payload,
                        const uint64_t *d_payload_offsets, const uint8_t *d_bits, uint32_t bit, int stream_layout,
                        const uint64_t *d_query_blocks, const uint32_t *d_keys, const int32_t *d_lengths,
                        size_t nqueries, int32_t *d_positions_out, uint32_t *d_values_out, uint32_t *d_last4_out,
                        void *stream);
/* scan: d_last4_out[4k..4k+3] = the last four values of block k -- what
 * simdscand1 (src/simdpackedselect.c:15216) leaves in its offset vector. */
int simdcomp_hip_scand1(uint32_t initvalue, const uint32_t *d_initvalues, const void *d_payload,
                        const uint64_t *d_payload_offsets, const uint8_t *d_bits, uint32_t bit, int stream_layout,
                        size_t nblocks, uint32_t *d_last4_out, void *stream);

/* A block directory: per block one 16-byte entry holding where its payload
 * starts, its width and its initvalue, so that a query on a block no
 * neighbouring query shares fetches one entry instead of three array elements
 * (2^24 selects with a random block each: 1.37 -> 1.06 ms, searches 1.60 ->
 * 1.54 ms; batches sorted by block gain 3 %).  No counterpart in the reference, whose callers keep
 * such a record per block themselves (example.c:101-110 walks the stream
 * instead).  Build it once per corpus from the arrays the other entry points
 * take (d_initvalues NULL: `initvalue` for every block; payload offsets must be
 * below 2^48), then query with select_directory / search_directory, which take
 * the directory in place of d_initvalues, d_payload_offsets, d_bits, bit and
 * stream_layout and answer exactly as simdcomp_hip_select / _search do.
 * d_directory: simdcomp_hip_block_directory_bytes(nblocks) bytes, 16-byte
 * aligned. */
size_t simdcomp_hip_block_directory_bytes(size_t nblocks);
int simdcomp_hip_block_directory(uint32_t initvalue, const uint32_t *d_initvalues, const uint64_t *d_payload_offsets,
                                 const uint8_t *d_bits, size_t nblocks, int stream_layout, void *d_directory, void *stream);
int simdcomp_hip_select_directory(int mode, const void *d_payload, const void *d_directory, const uint64_t *d_query_blocks,
                                  const int32_t *d_slots, size_t nqueries, uint32_t *d_values_out, void *stream);
int simdcomp_hip_search_directory(int mode, const void *d_payload, const void *d_directory, const uint64_t *d_query_blocks,
                                  const uint32_t *d_keys, const int32_t *d_lengths, size_t nqueries, int32_t *d_positions_out,
                                  uint32_t *d_values_out, uint32_t *d_last4_out, void *stream);

/* ---- the 8-lane (AVX2) and 16-lane (AVX-512) wire formats --------------------- */

/* Blocks of 32*lanes integers packed with `lanes` interleaved lanes: value i in
 * lane i % lanes, row i / lanes, a block is `bit` vectors of 4*lanes bytes --
 * what avxpack / avxpackwithoutmask / avxunpack (src/avxbitpacking.c:9906-9918,
 * lanes = 8, 256 integers) and avx512pack / avx512packwithoutmask / avx512unpack
 * (src/avx512bitpacking.c, lanes = 16, 512 integers) read and write, for
 * interoperability with data packed by those paths.  lanes = 4 is
 * simdcomp_hip_pack / _unpack.  nblocks counts blocks of 32*lanes integers;
 * plain coding at one uniform width only, as in the reference. */
int simdcomp_hip_pack_lanes(int lanes, const uint32_t *d_in, void *d_out, size_t nblocks, uint32_t bit, int masked,
                            void *stream);
int simdcomp_hip_unpack_lanes(int lanes, const void *d_in, uint32_t *d_out, size_t nblocks, uint32_t bit, void *stream);
/* avxmaxbits / avx512maxbits (src/avxbitpacking.c:17): width of the OR of the 32*lanes integers of each block. */
int simdcomp_hip_maxbits_lanes(int lanes, const uint32_t *d_in, size_t nblocks, uint8_t *d_bits, void *stream);

/* ---- block-range partitioner and verification reduction -------------------- */

/* Contiguous block range of shard `part` out of `nparts`: blocks are
 * independent (src/simdbitpacking.c:14382-14386 carries no state), so a corpus
 * is split by block index.  Host-side arithmetic only. */
int simdcomp_hip_partition(size_t nblocks, int nparts, int part, size_t *first, size_t *count);
/* Same, but balanced on payload+output bytes for mixed widths.  h_payload_offsets
 * is a HOST array of nblocks+1 byte offsets (last entry = total). */
int simdcomp_hip_partition_by_bytes(const uint64_t *h_payload_offsets, size_t nblocks, int nparts, int part,
                                    size_t *first, size_t *count);
/* d_acc[0] += position-sensitive 64-bit checksum of d_data[0..nwords) (word i
 * is hashed with global index index0+i, so shard sums add up to the checksum of
 * the whole array); if d_other is given, d_acc[1] += number of words that
 * differ from d_other.  d_acc is 2 x uint64 on the device, caller-zeroed. */
int simdcomp_hip_checksum(const uint32_t *d_data, const uint32_t *d_other, size_t nwords, uint64_t index0,
                          uint64_t *d_acc, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SIMDCOMP_HIP_H_ */
