// Kernels for the callers either side of the block codec (gfx950 / CDNA4).
//
//  * chained differential decode without stored offsets (example.c:101-110): the
//    offset of block k+1 is the last value of block k.  For b < 32 that is
//    offset[k] + (sum of the block's 128 deltas); a 32-bit block stores raw
//    values (src/simdintegratedbitpacking.c:24903-24913) and restarts the
//    chain at its own last value; a 0-bit block passes the offset through.
//    So offsets = an exclusive scan of (restart flag, value) pairs under the
//    associative operator  (f1,v1) . (f2,v2) = f2 ? (1,v2) : (f1, v1+v2).
//  * the byte stream of example.c:56-75: record k = [1 byte b][16*b bytes], which
//    puts record k at byte  k + 16*(b_0 + ... + b_{k-1})  and its payload at any
//    byte alignment.  build / unbuild move payloads between that stream and the
//    16-byte aligned arena the codec kernels use; index finds the records of a
//    stream when only its bytes are known.
#include "stream_kernels.h"

#include "lane_ops.h"

#ifndef SC_NT_SUMS
#define SC_NT_SUMS 0
#endif
#ifndef SC_NT_COPY
#define SC_NT_COPY 0
#endif

namespace sc {

// ---------------------------------------------------------------------------
// chained seeds
// ---------------------------------------------------------------------------
// sums[k] = sum of the 128 b-bit fields of block k (its deltas) modulo 2^32, or,
// for a 32-bit block, its last raw value.  One block per half-wave, rows
// fetched as in the decoder (wave-private LDS slot, funnel shift); nothing is
// stored but one word per block, so a wave takes U blocks per half with all
// loads issued up front to have enough bytes in flight.
//
// The fields are never isolated.  Let W_k be the 32 bits of a lane's bit stream
// that start at bit k*b (what the funnel shift delivers; zeros past the stream's
// end).  Then W_k = f_k + 2^b * W_{k+1} (mod 2^32), so with T = W_0 + ... + W_31
//     f_0 + ... + f_31 = T - 2^b * (T - W_0)   (mod 2^32),
// and the same holds for the sum over the block's four lane streams: no mask per
// value.  Lanes past a payload's end load 16 zero bytes.
//
// (Measured with it: 326 -> 224 vector instructions per wave, 6 % faster on the aligned arena, no change on the byte
// stream, where every payload load is unaligned.  Workgroups that touch widths and offsets ahead of the others, as
// in the per-block-width decoder, change nothing either way.)
#ifndef SC_SUMS_U
#define SC_SUMS_U 8
#endif
__constant__ u32x4 g_sixteen_zero_bytes = {0, 0, 0, 0};  // device code cannot write it; k_block_sums depends on its value
constexpr int SUMS_U = SC_SUMS_U;
constexpr size_t SUMS_PER_WG = 8 * SUMS_U;

// (Measured on 2^23 blocks of 7.5 bits per integer, 0.30 ms: a quarter fewer vector instructions -- the closing
// reduction below instead of a DPP reduction per block -- changed nothing, and workgroups that touch the payload
// ahead of the computing ones, as the uniform decoders have them, made it 8 % slower: neither the vector ALU nor the
// latency of the payload loads bounds this kernel.  What fits the time is the LDS data path, 128 bytes per cycle and
// CU: per wave 8 slot writes and 16 row reads of 1 KiB each, some 24 crossbar hand-overs and the shares, about 320
// cycles, times 2048 waves per CU = 0.27 ms.)
__global__ __launch_bounds__(256) void k_block_sums(const uint8_t *payload, const uint64_t *off, const uint8_t *bits,
                                                    size_t nblocks, uint32_t stream_layout, uint32_t *sums) {
  constexpr int U = SUMS_U;
  const size_t tile = blockIdx.x;
  static_assert(U == 8, "the closing reduction spreads 2 * U = 16 (half-wave, step) pairs over the 64 lanes, four lanes each");
  __shared__ u32x4 lds[4][2][33];
  // every lane's share of each of its blocks, [half-wave][step][row]: summed over the rows at the end, all blocks of
  // the wave at once (a reduction across the lanes per step costs five dependent DPP additions and their wait states
  // per block; this way the wave pays seven plain additions and two DPP steps for all sixteen)
  __shared__ uint32_t shares[4][2][U][32];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  u32x4 *slot = lds[wv][h];
  uint32_t b[U];
  uintptr_t at[U];
  u32x4 ld[U];
  uintptr_t first_payload;
  uint32_t mb;
  // widths and offsets of the wave's 2 * U consecutive blocks: one coalesced load per array (blocks past the end
  // repeat the last one), handed to the half-waves through the LDS crossbar (a ds_bpermute costs one instruction
  // where the scalar hand-out of the decoders costs four).  The offsets travel as 32-bit distances from the wave's
  // first block (with offsets from a scan 16 blocks span at most 8 KiB); when a block of the wave lies below the
  // first one or 4 GiB and more above it (payload offsets may be any multiples of 16) the high halves travel too.
  constexpr int NB = 2 * U;
  const size_t wb = (tile * 4 + wv) * NB;
  if (wb >= nblocks) return;  // (whole waves; no workgroup barrier below)
  {
    const size_t mine = wb + (size_t)(lane % NB) < nblocks ? wb + (size_t)(lane % NB) : nblocks - 1;
    mb = bits[mine];
    mb = mb > 32 ? 32 : mb;
    const uint64_t mo = off[mine] + (stream_layout ? mine + 1 : 0);
    const uint64_t mo0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mo >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mo);
    first_payload = reinterpret_cast<uintptr_t>(payload) + mo0;
    const uint64_t dist = mo - mo0;
    const uint32_t rel = (uint32_t)dist, rel_hi = (uint32_t)(dist >> 32);
    const bool any_far = __builtin_amdgcn_ballot_w64(rel_hi != 0) != 0;  // wave-uniform; no load sits behind this branch
    uint32_t o[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int from = 4 * (2 * u + h);  // byte address of the source lane
      b[u] = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)mb);
      o[u] = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)rel);
    }
    if (!any_far) {
#pragma unroll
      for (int u = 0; u < U; ++u) at[u] = first_payload + (o[u] + 16 * (uint32_t)k);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (2 * u + h), (int)rel_hi);
        at[u] = first_payload + (((uint64_t)hi << 32) | o[u]) + 16 * (uint64_t)k;
      }
    }
  }
  // lanes past a payload's end must contribute zeros (the sums below rely on it): they load sixteen zero bytes from
  // read-only memory of this kernel's own (not the scratch vector other kernels use as a dummy load target)
  const uintptr_t zeros = reinterpret_cast<uintptr_t>(&g_sixteen_zero_bytes);
#pragma unroll
  for (int u = 0; u < U; ++u)
    ld[u] = load_global_at((uint32_t)k < b[u] ? at[u] : zeros);
  uint32_t *my_share = &shares[wv][h][0][k];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    wave_lds_fence();
    slot[k] = ld[u];
    wave_lds_fence();
    const uint32_t p = __umul24((uint32_t)k, b[u]), w = p >> 5, s = p & 31;
    const u32x4 lo = slot[w], hi = slot[w + 1];  // b = 32: s = 0, hi (slot[32] for the last row) is not used
    const uint32_t x = funnel(hi.x, lo.x, s), y = funnel(hi.y, lo.y, s), z = funnel(hi.z, lo.z, s), q = funnel(hi.w, lo.w, s);
    // row 0's share is W_0 (s = 0, w = 0); a 32-bit block only needs its last raw value, row 31's q
    my_share[32 * u] = b[u] == 32 ? q : x + y + z + q;
  }
  wave_lds_fence();
  // lane 4 i + part adds rows 8 part .. 8 part + 7 of the wave's i-th (half-wave, step) pair; two DPP steps bring the
  // four parts together in the quad's last lane, which also holds row 31's share; row 0's comes from the quad's first
  {
    const int i = lane >> 2, part = lane & 3;
    const int hh = i / U, uu = i % U;
    const u32x4 *eight = reinterpret_cast<const u32x4 *>(&shares[wv][hh][uu][8 * part]);
    const u32x4 a4 = eight[0], b4 = eight[1];
    uint32_t t = (a4.x + a4.y + a4.z) + (a4.w + b4.x + b4.y) + (b4.z + b4.w);
    t += dpp0<DPP_ROW_SHR1>(t) & ((part & 1) ? 0xFFFFFFFFu : 0u);  // parts 1 and 3 take their neighbour's sum
    const uint32_t pair = dpp0<DPP_ROW_SHR2>(t);                    // part 3 takes parts 0 + 1
    t += pair;
    const uint32_t w0 = dpp0<0x113>(a4.x);  // row_shr:3: the first row's share, from the quad's first lane
    const size_t blk = wb + (size_t)(2 * uu + hh);
    const uint32_t bb = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (2 * uu + hh), (int)mb);
    if (part == 3 && blk < nblocks) sums[blk] = bb >= 32 ? b4.w : t - ((t - w0) << bb);
  }
}

// (restart flag << 32) | value
__device__ __forceinline__ uint64_t chain_compose(uint64_t a, uint64_t b) {
  return (b >> 32) ? b : ((a & 0xFFFFFFFF00000000ull) | (uint32_t)((uint32_t)a + (uint32_t)b));
}
__device__ __forceinline__ uint64_t chain_element(uint32_t width, uint32_t sum) {
  return ((uint64_t)(width >= 32) << 32) | sum;
}

constexpr int CHAIN_TILE = 4096;  // blocks per workgroup: 256 threads x 16

// Inclusive scan over the 256 threads of a workgroup under chain_compose; leaves the result in sh[].
__device__ __forceinline__ void chain_wg_scan(uint64_t v, uint64_t *sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const uint64_t t = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] = chain_compose(t, sh[threadIdx.x]);
    __syncthreads();
  }
}

// A tile is CHAIN_TILE blocks: wave w takes 1024 consecutive ones in 16 rounds of 64, lane l the block at
// 64 * round + l: every load and store is one coalesced instruction, and the 32 loads go out back to back (blocks past
// the end re-read the last one and count as the identity).
__device__ __forceinline__ void chain_load(const uint8_t *bits, const uint32_t *sums, size_t n, size_t wave_base, int lane,
                                           uint64_t (&el)[16]) {
  uint32_t w[16], sm[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const size_t i = wave_base + 64 * (size_t)r + (size_t)lane;
    const size_t ic = i < n ? i : n - 1;
    w[r] = bits[ic];
    sm[r] = sums[ic];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) el[r] = wave_base + 64 * (size_t)r + (size_t)lane < n ? chain_element(w[r], sm[r]) : 0;
}

// Scans the wave's 16 rounds in place (el[r] becomes the inclusive scan within the round) and returns what the wave's
// 1024 blocks compose to; before[r] is what the rounds in front of round r compose to.
__device__ __forceinline__ uint64_t chain_wave_scan(uint64_t (&el)[16], uint64_t (&before)[16], int lane) {
  uint64_t run = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    el[r] = wave_scan_incl(el[r], lane, chain_compose);
    before[r] = run;
    run = chain_compose(run, __shfl((unsigned long long)el[r], 63, 64));
  }
  return run;
}

__global__ __launch_bounds__(256) void k_chain_tiles(const uint8_t *bits, const uint32_t *sums, size_t n, uint64_t *tile_sum) {
  __shared__ uint64_t sh[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint64_t el[16], before[16];
  chain_load(bits, sums, n, (size_t)blockIdx.x * CHAIN_TILE + (size_t)wv * 1024, lane, el);
  const uint64_t total = chain_wave_scan(el, before, lane);
  if (lane == 0) sh[wv] = total;
  __syncthreads();
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = chain_compose(chain_compose(sh[0], sh[1]), chain_compose(sh[2], sh[3]));
}

// one workgroup: tile_in[t] = offset entering tile t
__global__ __launch_bounds__(256) void k_chain_spine(const uint64_t *tile_sum, size_t ntiles, uint32_t init, uint32_t *tile_in) {
  __shared__ uint64_t sh[256];
  uint64_t carry = (1ull << 32) | init;
  for (size_t base = 0; base < ntiles; base += 256) {
    const size_t i = base + threadIdx.x;
    chain_wg_scan(i < ntiles ? tile_sum[i] : 0, sh);
    const uint64_t before = threadIdx.x ? sh[threadIdx.x - 1] : 0;
    if (i < ntiles) tile_in[i] = (uint32_t)chain_compose(carry, before);
    carry = chain_compose(carry, sh[255]);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_chain_apply(const uint8_t *bits, const uint32_t *sums, size_t n, const uint32_t *tile_in,
                                                     uint32_t *seeds) {
  __shared__ uint64_t sh[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t wave_base = (size_t)blockIdx.x * CHAIN_TILE + (size_t)wv * 1024;
  uint64_t el[16], before[16];
  chain_load(bits, sums, n, wave_base, lane, el);
  const uint64_t entering = (1ull << 32) | tile_in[blockIdx.x];
  const uint64_t total = chain_wave_scan(el, before, lane);
  if (lane == 0) sh[wv] = total;
  __syncthreads();
  uint64_t state = entering;
  for (int w = 0; w < wv; ++w) state = chain_compose(state, sh[w]);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    // the offset of a block: everything in front of it, i.e. the inclusive scan of the lane before
    const uint64_t prev = __shfl_up((unsigned long long)el[r], 1, 64);
    const uint64_t in_round = lane ? prev : 0;
    const size_t i = wave_base + 64 * (size_t)r + (size_t)lane;
    if (i < n) seeds[i] = (uint32_t)chain_compose(chain_compose(state, before[r]), in_round);
  }
}

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

size_t chain_workspace_bytes(size_t nblocks) {
  const size_t ntiles = (nblocks + CHAIN_TILE - 1) / CHAIN_TILE;
  return align256(4 * nblocks) + align256(8 * ntiles) + align256(4 * ntiles) + 256;
}

hipError_t launch_chain_offsets(uint32_t init, const void *payload, const uint64_t *off, const uint8_t *bits, size_t nblocks,
                                bool stream_layout, uint32_t *seeds_out, void *workspace, hipStream_t st) {
  if (nblocks == 0) return hipSuccess;
  const size_t ntiles = (nblocks + CHAIN_TILE - 1) / CHAIN_TILE;
  uint8_t *ws = static_cast<uint8_t *>(workspace);
  uint32_t *sums = reinterpret_cast<uint32_t *>(ws);
  uint64_t *tile_sum = reinterpret_cast<uint64_t *>(ws + align256(4 * nblocks));
  uint32_t *tile_in = reinterpret_cast<uint32_t *>(ws + align256(4 * nblocks) + align256(8 * ntiles));
  const size_t nwg = (nblocks + SUMS_PER_WG - 1) / SUMS_PER_WG;
  hipLaunchKernelGGL(k_block_sums, dim3((unsigned)nwg), dim3(256), 0, st,
                     static_cast<const uint8_t *>(payload), off, bits, nblocks, stream_layout ? 1u : 0u, sums);
  hipLaunchKernelGGL(k_chain_tiles, dim3((unsigned)ntiles), dim3(256), 0, st, bits, sums, nblocks, tile_sum);
  hipLaunchKernelGGL(k_chain_spine, dim3(1), dim3(256), 0, st, tile_sum, ntiles, init, tile_in);
  hipLaunchKernelGGL(k_chain_apply, dim3((unsigned)ntiles), dim3(256), 0, st, bits, sums, nblocks, tile_in, seeds_out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// [b][payload] stream <-> aligned arena
// ---------------------------------------------------------------------------
// A wave takes 64 consecutive blocks, lane l the width and offset of block l, and moves their payload vectors
// densely: vector i of the wave's V = b_0 + ... + b_63 vectors goes to lane i mod 64 of round i / 64, so every load
// and store instruction moves 64 vectors whatever the widths are.  (With one block per half-wave and lane k < b
// moving vector k, as this kernel worked before, a quarter of the lanes were busy at 7.5 bits per integer and the
// copy was bound by the rate of memory instructions: 16 cycles of the texture addresser per 16-byte instruction of
// a wave, busy lanes or not.)  The owner of vector i -- the block l with P_l <= i < P_l + b_l, P the exclusive scan of
// the widths over the wave -- comes from a table in LDS: block l writes l + 1 at entry P_l, a running maximum over
// the entries fills in the rest (one DPP scan per round, carried from round to round).  The owner's P and offset
// travel through the LDS crossbar.  Loads of COPY_G rounds go out together, stores after; the stream side is at
// an arbitrary byte address.  TO_STREAM also writes the width bytes.
#ifndef SC_COPY_G
#define SC_COPY_G 8
#endif
constexpr int COPY_G = SC_COPY_G;
constexpr int COPY_WAVE_BLOCKS = 64;

__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t t) {
  t += dpp0<DPP_ROW_SHR1>(t);
  t += dpp0<DPP_ROW_SHR2>(t);
  t += dpp0<DPP_ROW_SHR4>(t);
  t += dpp0<DPP_ROW_SHR8>(t);
  t += dpp0<DPP_ROW_BCAST15, 0xA>(t);
  t += dpp0<DPP_ROW_BCAST31, 0xC>(t);
  return t;
}
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wave_scan_max_u32(uint32_t t) {
  t = umax(t, dpp0<DPP_ROW_SHR1>(t));
  t = umax(t, dpp0<DPP_ROW_SHR2>(t));
  t = umax(t, dpp0<DPP_ROW_SHR4>(t));
  t = umax(t, dpp0<DPP_ROW_SHR8>(t));
  t = umax(t, dpp0<DPP_ROW_BCAST15, 0xA>(t));
  t = umax(t, dpp0<DPP_ROW_BCAST31, 0xC>(t));
  return t;
}

template <bool TO_STREAM>
__global__ __launch_bounds__(256) void k_stream_copy(const uint8_t *src, uint8_t *dst, const uint64_t *off, const uint8_t *bits,
                                                     size_t nblocks) {
  constexpr int G = COPY_G;
  constexpr int TABLE = 32 * COPY_WAVE_BLOCKS;  // most vectors 64 blocks can have
  __shared__ u32x4 owner_mem[4][TABLE / 16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t wb = ((size_t)blockIdx.x * 4 + wv) * COPY_WAVE_BLOCKS;
  if (wb >= nblocks) return;  // (whole waves; there is no workgroup barrier in this kernel)
  const size_t blk = wb + (size_t)lane;
  const bool have = blk < nblocks;
  const size_t bc = have ? blk : nblocks - 1;
  uint32_t b = bits[bc];
  const uint64_t o = off[bc];
  b = b > 32 ? 32 : b;
  b = have ? b : 0;
  const uint32_t incl = wave_scan_add_u32(b);
  const uint32_t P = incl - b;
  const uint32_t V = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  uint8_t *owner = reinterpret_cast<uint8_t *>(owner_mem[wv]);
  owner_mem[wv][lane] = u32x4{0, 0, 0, 0};
  owner_mem[wv][lane + 64] = u32x4{0, 0, 0, 0};
  wave_lds_fence();
  if (b) owner[P] = (uint8_t)(lane + 1);
  wave_lds_fence();
  uint32_t carry = 0;
  for (uint32_t r0 = 0; r0 * 64 < V; r0 += G) {
    u32x4 x[G];
    uint64_t to[G];
    bool act[G], head[G];
    uint32_t wbyte[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const uint32_t i = (r0 + g) * 64 + (uint32_t)lane;
      uint32_t ow = owner[i < TABLE ? i : TABLE - 1];
      ow = umax(wave_scan_max_u32(ow), carry);
      carry = (uint32_t)__builtin_amdgcn_readlane((int)ow, 63);
      act[g] = i < V;
      const int l4 = (int)(ow ? ow - 1 : 0) * 4;  // the owner's lane, as the byte address the crossbar wants
      const uint32_t Pl = (uint32_t)__builtin_amdgcn_ds_bpermute(l4, (int)P);
      const uint64_t ol = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(l4, (int)(uint32_t)(o >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_ds_bpermute(l4, (int)(uint32_t)o);
      const uint64_t in_arena = ol + 16 * (uint64_t)(i - Pl);
      const uint64_t in_stream = in_arena + wb + (uint64_t)(l4 >> 2) + 1;
      to[g] = TO_STREAM ? in_stream : in_arena;
      const uintptr_t from = reinterpret_cast<uintptr_t>(src) + (TO_STREAM ? in_arena : in_stream);
      // lanes past the last vector read 16 bytes that are always there: a predicated load would sit behind a branch
      x[g] = SC_NT_COPY ? load_global_nt(act[g] ? from : reinterpret_cast<uintptr_t>(&g_nothing))
                        : load_global_at(act[g] ? from : reinterpret_cast<uintptr_t>(&g_nothing));
      if (TO_STREAM) {
        wbyte[g] = (uint32_t)__builtin_amdgcn_ds_bpermute(l4, (int)b);
        head[g] = act[g] && i == Pl;
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      // the width byte goes out next to the record's first vector: the two share a line
      if (TO_STREAM && head[g]) dst[to[g] - 1] = (uint8_t)wbyte[g];
      if (act[g]) {
        if (SC_NT_COPY) __builtin_nontemporal_store(x[g], reinterpret_cast<u32x4_u *>(dst + to[g]));
        else *reinterpret_cast<u32x4_u *>(dst + to[g]) = x[g];
      }
    }
  }
  // zero-width blocks have no vector to travel with (behind the payload: loads and stores share one in-order
  // counter, so a store issued up front would make the first wait for a load wait for its acknowledgement too)
  if (TO_STREAM && have && b == 0) dst[o + blk] = 0;
}

hipError_t launch_stream_build(const void *arena, const uint64_t *off, const uint8_t *bits, size_t nblocks, uint8_t *stream,
                               hipStream_t st) {
  if (nblocks == 0) return hipSuccess;
  constexpr size_t per_wg = 4 * COPY_WAVE_BLOCKS;
  hipLaunchKernelGGL(k_stream_copy<true>, dim3((unsigned)((nblocks + per_wg - 1) / per_wg)), dim3(256), 0, st,
                     static_cast<const uint8_t *>(arena), stream, off, bits, nblocks);
  return hipGetLastError();
}

hipError_t launch_stream_unbuild(const uint8_t *stream, const uint64_t *off, const uint8_t *bits, size_t nblocks, void *arena,
                                 hipStream_t st) {
  if (nblocks == 0) return hipSuccess;
  constexpr size_t per_wg = 4 * COPY_WAVE_BLOCKS;
  hipLaunchKernelGGL(k_stream_copy<false>, dim3((unsigned)((nblocks + per_wg - 1) / per_wg)), dim3(256), 0, st, stream,
                     static_cast<uint8_t *>(arena), off, bits, nblocks);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// stream index: where do the records start?
// ---------------------------------------------------------------------------
// Record k+1 starts 1 + 16*b_k bytes after record k: a chain through the data.  It is cut into chunks of IDX_CHUNK
// bytes.  The first record of a chunk starts at one of 513 places (a record is at most 513 bytes long), so
//   A  every chunk tabulates, per candidate, where the walk from it leaves the chunk (0..512 bytes into the next
//      one) and how many records it sees: the candidates are resolved against each other inside the first 1.5 KiB of
//      the chunk (kernel A, that part staged in LDS), the few walks that are still alive behind it are finished from
//      global memory, every chunk of the stream at the same time (kernel A'), logging the widths they read;
//   B  groups of IDX_GROUP chunks compose their tables (thread f follows candidate f through the group), and
//      super-groups of IDX_SUPER groups compose the group tables the same way: a chain of 32 dependent loads each,
//      instead of one long chain;
//   C  one thread walks the super-group tables from the known entry of the stream (byte 0) and fixes every
//      super-group's entry and first record number;  D  one thread per super-group does the same for its groups,
//      then one thread per group for its chunks;
//   E  every chunk writes width and arena offset of each record of its true chain: the few records in front of the
//      logged walk by walking, the rest from the log.
// (Staging whole chunks in LDS for A and walking the true chain again in E, as this code did at first, cost 0.58 ms
// and 0.22 ms per GiB of stream; A + A' take 0.13 + 0.25 ms, E 0.065 ms.)
// A width byte above 32 ends a walk as invalid; the true chain hitting one is reported in result[2].
#ifndef SC_IDX_CHUNK
#define SC_IDX_CHUNK 16384
#endif
constexpr int IDX_CHUNK = SC_IDX_CHUNK;
constexpr int IDX_CAND = 513;
constexpr int IDX_GROUP = 32;  // chunks per group
constexpr int IDX_SUPER = 32;  // groups per super-group
constexpr uint32_t IDX_INVALID = 0xFFFFFFFFu;
constexpr uint64_t IDX_INVALID64 = ~0ull;

// Kernel A.  Candidate f's walk goes through byte f + 1 + 16*b(f); when that is again one of the 513 candidates the
// rest of the walk is that candidate's, so only "roots" (candidates whose first step already leaves the candidate
// window) walk on; the others are linked to their successor and resolved by ten rounds of pointer jumping over the
// 513 entries.  Only the first 513 * IDX_PROBE bytes of the chunk are staged in LDS: a root walks until it crosses
// that boundary, which is where nearly all false candidates die (a payload byte is a valid width byte several times
// in a row only rarely) and where walks that have met the true chain on the way arrive at the same byte.  A root that
// crosses the boundary is a "survivor": survivors arriving at the same byte share one of the chunk's IDX_SLOTS slots,
// kernel A' walks each slot to the end of the chunk (one lane per slot, all chunks of the stream at once), and the
// table keeps (records so far, slot) for the candidates that end in a slot.
// res[] and table words: (records << 16) | target; target < 0x4000: done, the walk left the chunk `target` bytes into
// the next one; target = 0x4000 + f: continue at candidate f (inside kernel A only); target = 0x8000 + s: continues in
// slot s.
constexpr uint32_t IDX_LINK = 0x4000;
constexpr uint32_t IDX_PENDING = 0x8000;
#ifndef SC_IDX_SLOTS
#define SC_IDX_SLOTS 8
#endif
#ifndef SC_IDX_PROBE
#define SC_IDX_PROBE 3
#endif
constexpr int IDX_SLOTS = SC_IDX_SLOTS;
constexpr int IDX_PROBE = SC_IDX_PROBE;
constexpr int CAND_THREADS = 64;

__global__ __launch_bounds__(CAND_THREADS) void k_index_candidates(const uint8_t *stream, size_t nbytes, uint32_t *table,
                                                                   uint32_t *slot_start, uint32_t *pool_next) {
  constexpr uint32_t BOUNDARY = IDX_CAND * IDX_PROBE;  // roots walk until they cross this byte
  constexpr uint32_t WINDOW = BOUNDARY;                // so this much of the chunk is staged
  // (the workgroup is one wave: a wave-level fence is all the synchronisation the LDS arrays need.  The kernel is
  // bound by instruction issue, about a thousand instructions per chunk; several chunks per workgroup, one per
  // wave, measured the same.)
  __shared__ u32x4 window[(WINDOW + 15) / 16];
  __shared__ __attribute__((aligned(16))) uint32_t res[IDX_CAND + 3];
  __shared__ uint16_t lists[IDX_CAND + 1];  // roots from the front, links from the back (a candidate is one or neither)
  const uint32_t lane = threadIdx.x;
  const size_t chunk = blockIdx.x;
  if (chunk == 0 && lane == 0) *pool_next = 0;  // the survivors kernel, which runs after this one, hands out logs from 0
  const size_t c0 = chunk * IDX_CHUNK;
  const uint8_t *src = stream + c0;
  const uint32_t len = (uint32_t)(nbytes - c0 < (size_t)IDX_CHUNK ? nbytes - c0 : (size_t)IDX_CHUNK);
  const uint32_t win = len < WINDOW ? len : WINDOW;
  uint8_t *by = reinterpret_cast<uint8_t *>(window);
  {
    // all loads of the lane first, then the LDS writes (one global round trip, not one per 1 KiB)
    constexpr int VECS = ((WINDOW + 15) / 16 + CAND_THREADS - 1) / CAND_THREADS;
    u32x4 v[VECS];
    uint8_t tail = 0;
    const uint32_t j = (win & ~15u) + lane;  // (at most 15 bytes behind the last whole vector)
#pragma unroll
    for (int k = 0; k < VECS; ++k) {
      const uint32_t i = lane + k * CAND_THREADS;
      if (16 * (i + 1) <= win) v[k] = *reinterpret_cast<const u32x4_u *>(src + 16 * i);
    }
    if (j < win) tail = src[j];
#pragma unroll
    for (int k = 0; k < VECS; ++k) {
      const uint32_t i = lane + k * CAND_THREADS;
      if (16 * (i + 1) <= win) window[i] = v[k];
    }
    if (j < win) by[j] = tail;
  }
  wave_lds_fence();
  // every lane counts the roots and links among its own candidates (bit j of rootbits / linkbits: its j-th one), one
  // prefix sum over the wave gives it its places in the two lists (both counts in one word)
  constexpr int PER_LANE = (IDX_CAND + CAND_THREADS - 1) / CAND_THREADS;
  uint32_t rootbits = 0, linkbits = 0;
  const bool whole = len >= 2 * (uint32_t)IDX_CAND;  // no first step leaves the chunk: all but the last chunk of a stream
  if (whole) {
    // lane l takes candidates 8 l .. 8 l + 7 (and lane 0 the 513th).  Seven of eight bytes are no width bytes: all
    // eight are sorted out at once (bit 7 of a byte of `over` is set when the byte is above 32), the entries are
    // written as invalid with two 16-byte stores, and only the valid ones are looked at one by one.
    const uint64_t x = *reinterpret_cast<const uint64_t *>(by + 8 * lane);
    auto valid_nibble = [](uint32_t w) {
      const uint32_t over = (((w & 0x7F7F7F7Fu) + 0x5F5F5F5Fu) | w) & 0x80808080u;
      return ((((~over) >> 7) & 0x01010101u) * 0x01020408u) >> 24;  // bits 7, 15, 23, 31 gathered into bits 0..3
    };
    uint32_t valid = valid_nibble((uint32_t)x) | (valid_nibble((uint32_t)(x >> 32)) << 4);
    const u32x4 none = {IDX_INVALID, IDX_INVALID, IDX_INVALID, IDX_INVALID};
    reinterpret_cast<u32x4 *>(res)[2 * lane] = none;
    reinterpret_cast<u32x4 *>(res)[2 * lane + 1] = none;
    while (valid) {
      const uint32_t j = (uint32_t)__ffs((int)valid) - 1;
      valid &= valid - 1;
      const uint32_t f = 8 * lane + j, nf = f + 1 + 16 * ((uint32_t)(x >> (8 * j)) & 0xFF);
      if (nf < (uint32_t)IDX_CAND) {
        res[f] = (1u << 16) | IDX_LINK | nf;
        linkbits |= 1u << j;
      } else {
        rootbits |= 1u << j;  // res[f] comes from the walk below
      }
    }
    if (lane == 0) {
      res[IDX_CAND - 1] = IDX_INVALID;
      if (by[IDX_CAND - 1] <= 32) rootbits |= 1u << 8;
    }
  } else {
    uint32_t width[PER_LANE];
#pragma unroll
    for (int k = 0; k < PER_LANE; ++k) {
      const uint32_t f = lane + k * CAND_THREADS;
      width[k] = f < IDX_CAND && f < len ? by[f] : 0xFF;
    }
#pragma unroll
    for (int k = 0; k < PER_LANE; ++k) {  // lane l takes candidates l, l + 64, ...
      const uint32_t f = lane + k * CAND_THREADS;
      const uint32_t b = width[k];  // 0xFF outside the chunk or the window
      const uint32_t nf = f + 1 + 16 * b;
      const bool valid = b <= 32, leaves = nf >= len;
      const bool link = valid && !leaves && nf < (uint32_t)IDX_CAND;
      const bool root = valid && !leaves && nf >= (uint32_t)IDX_CAND;  // res[f] comes from the walk below
      uint32_t word = IDX_INVALID;
      word = valid && leaves ? ((1u << 16) | (nf - len)) : word;
      word = link ? ((1u << 16) | IDX_LINK | nf) : word;
      word = f >= len ? f - len : word;  // the chunk is shorter than the candidate window
      linkbits |= (uint32_t)link << k;
      rootbits |= (uint32_t)root << k;
      if (f < (uint32_t)IDX_CAND) res[f] = word;
    }
  }
  auto candidate = [&](int j) -> uint32_t {  // the lane's j-th candidate
    return whole ? (j == 8 ? (uint32_t)IDX_CAND - 1 : 8 * lane + (uint32_t)j) : lane + (uint32_t)j * CAND_THREADS;
  };
  const uint32_t mine = (uint32_t)__popc(rootbits) | ((uint32_t)__popc(linkbits) << 16);
  uint32_t incl = mine;
  incl += dpp0<DPP_ROW_SHR1>(incl);
  incl += dpp0<DPP_ROW_SHR2>(incl);
  incl += dpp0<DPP_ROW_SHR4>(incl);
  incl += dpp0<DPP_ROW_SHR8>(incl);
  incl += dpp0<DPP_ROW_BCAST15, 0xA>(incl);
  incl += dpp0<DPP_ROW_BCAST31, 0xC>(incl);
  const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  const uint32_t total = all & 0xFFFF;
  uint32_t nlinks = all >> 16;
  uint32_t root_at = (incl - mine) & 0xFFFF, link_at = (incl - mine) >> 16;
  while (rootbits) {
    const int k = __ffs((int)rootbits) - 1;
    rootbits &= rootbits - 1;
    lists[root_at++] = (uint16_t)candidate(k);
  }
  while (linkbits) {
    const int k = __ffs((int)linkbits) - 1;
    linkbits &= linkbits - 1;
    lists[IDX_CAND - link_at++] = (uint16_t)candidate(k);
  }
  wave_lds_fence();
  uint32_t slot_pos[IDX_SLOTS];
  uint32_t nslots = 0;  // both the same in every lane
#pragma unroll
  for (int k = 0; k < IDX_SLOTS; ++k) slot_pos[k] = IDX_INVALID;
  for (uint32_t base = 0; base < total; base += CAND_THREADS) {
    const uint32_t t = base + lane;
    const uint32_t r = t < total ? lists[t] : 0;
    uint32_t p = r + 1 + 16 * (uint32_t)by[r], cnt = 1;
    bool walking = t < total, dead = false;
    while (walking && p < BOUNDARY) {
      const uint32_t bb = by[p];
      if (bb > 32) {
        dead = true;
        walking = false;
      } else {
        p += 1 + 16 * bb;
        ++cnt;
        walking = p < len;
      }
    }
    // survivors standing on the same byte have the same future: one slot for all of them
    int slot = -1;
    uint64_t todo = __ballot(walking);
    while (todo) {
      const int leader = __ffsll((unsigned long long)todo) - 1;
      const uint32_t lp = __shfl(p, leader);
      int s = -1;
#pragma unroll
      for (int k = 0; k < IDX_SLOTS; ++k)
        if (slot_pos[k] == lp) s = k;
      if (s < 0 && nslots < (uint32_t)IDX_SLOTS) {
        s = (int)nslots++;
#pragma unroll
        for (int k = 0; k < IDX_SLOTS; ++k)
          if (k == s) slot_pos[k] = lp;
      }
      const bool same = walking && p == lp;
      if (same) slot = s;
      todo &= ~__ballot(same);
    }
    if (walking && slot < 0) {
      // more distinct survivors than slots (streams made to look like that): this one walks to the end right here
      while (p < len) {
        const uint32_t bb = src[p];
        if (bb > 32) {
          dead = true;
          break;
        }
        p += 1 + 16 * bb;
        ++cnt;
      }
      walking = false;
    }
    if (t < total)
      res[r] = dead ? IDX_INVALID : ((cnt << 16) | (walking ? IDX_PENDING + (uint32_t)slot : p - len));
  }
  if (lane < IDX_SLOTS) {
    uint32_t v = IDX_INVALID;
#pragma unroll
    for (int k = 0; k < IDX_SLOTS; ++k)
      if ((int)lane == k) v = slot_pos[k];
    slot_start[chunk * IDX_SLOTS + lane] = v;
  }
  wave_lds_fence();
  const uint64_t below = (1ull << lane) - 1;
  // pointer jumping over the links only; a link whose target is resolved is resolved itself and leaves the list
  for (int round = 0; round < 10 && nlinks; ++round) {  // 2^10 >= the longest chain of links (513)
    uint32_t kept = 0;
    for (uint32_t base = 0; base < nlinks; base += CAND_THREADS) {
      const uint32_t t = base + lane;
      const bool have = t < nlinks;
      const uint32_t f = have ? lists[IDX_CAND - t] : 0;
      uint32_t w = have ? res[f] : (1u << 16);
      const uint32_t tw = res[w & 0x3FF];
      const bool dead = tw == IDX_INVALID;
      w = dead ? IDX_INVALID : ((((w >> 16) + (tw >> 16)) << 16) | (tw & 0xFFFF));
      const bool still = have && !dead && (tw & IDX_LINK);
      const uint64_t smask = __ballot(still);
      wave_lds_fence();  // every lane has read res[] and links[] of this batch (and of the rounds before)
      if (have) res[f] = w;
      if (still) lists[IDX_CAND - (kept + __popcll(smask & below))] = (uint16_t)f;
      kept += __popcll(smask);
    }
    wave_lds_fence();
    nlinks = kept;
  }
  uint32_t out[PER_LANE];
#pragma unroll
  for (int k = 0; k < PER_LANE; ++k) {
    const uint32_t f = lane + k * CAND_THREADS;
    out[k] = f < IDX_CAND ? res[f] : IDX_INVALID;
  }
#pragma unroll
  for (int k = 0; k < PER_LANE; ++k) {
    const uint32_t f = lane + k * CAND_THREADS;
    if (f < IDX_CAND) table[chunk * IDX_CAND + f] = out[k];
  }
}

// Kernel A'.  One lane per slot: the rest of the survivor's walk, from global memory.  slot_result: (records from
// the slot's first byte on << 16) | bytes into the next chunk, or invalid.  The walk also leaves the width bytes it
// read in a log (IDX_LOG records; a slot takes its log from a pool with a wave-wide atomic), so that the kernel that
// writes the index does not walk the true chain a second time.
constexpr uint32_t IDX_LOG = 1024;  // a 16 KiB chunk holds at most 964 records of one bit or more

static size_t idx_log_regions(size_t nchunks) { return nchunks + nchunks / 2 + 64; }

constexpr int SURV_THREADS = 1024;

__global__ __launch_bounds__(SURV_THREADS) void k_index_survivors(const uint8_t *stream, size_t nbytes, size_t nslots,
                                                                  const uint32_t *slot_start, uint32_t *slot_result,
                                                                  uint32_t *slot_log, uint32_t *pool_next, uint32_t nregions,
                                                                  uint8_t *logs) {
  // Most slots are empty (a chunk has IDX_SLOTS of them and typically uses one): the workgroup looks at 1024 slots,
  // gathers the live ones into its first lanes and lets the other waves go, so that the walk below runs in full
  // waves (with the live slots scattered over all waves, at a quarter of the lanes, it ran 15 % slower, and slower
  // still with more slots per chunk).
  __shared__ uint16_t live[SURV_THREADS];
  __shared__ uint32_t wave_count[SURV_THREADS / 64];
  const uint32_t wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  const size_t first_id = (size_t)blockIdx.x * SURV_THREADS;
  {
    const size_t id = first_id + threadIdx.x;
    const bool have = id < nslots && slot_start[id] != IDX_INVALID;
    const uint64_t mask = __ballot(have);
    if (lane == 0) wave_count[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += wave_count[w];
    if (have) live[before + (uint32_t)__popcll(mask & ((1ull << lane) - 1))] = (uint16_t)threadIdx.x;
    __syncthreads();
  }
  uint32_t nlive = 0;
#pragma unroll
  for (int w = 0; w < SURV_THREADS / 64; ++w) nlive += wave_count[w];
  if (threadIdx.x >= nlive) return;  // (whole waves leave here, and the tail of the last live one)
  const size_t id = first_id + live[threadIdx.x];
  const uint32_t start = slot_start[id];
  // a log for every lane that is left: one atomic per wave
  const uint32_t mine = nlive - wave * 64 < 64 ? nlive - wave * 64 : 64;
  uint32_t first = 0;
  if (lane == 0) first = atomicAdd(pool_next, mine);
  first = __shfl(first, 0);
  uint32_t region = first + lane;
  if (region >= nregions) region = IDX_INVALID;  // (streams made to have many survivors: the index is written by a walk)
  slot_log[id] = region;
  const size_t c0 = id / IDX_SLOTS * IDX_CHUNK;
  const uint8_t *src = stream + c0;
  const uint32_t len = (uint32_t)(nbytes - c0 < (size_t)IDX_CHUNK ? nbytes - c0 : (size_t)IDX_CHUNK);
  uint32_t p = start, cnt = 0;
  bool bad = false;
  typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
  u64x2 *log = reinterpret_cast<u64x2 *>(logs + (size_t)region * IDX_LOG);
  while (p < len && !bad) {
    // sixteen dependent loads, then one store (memory operations of a wave complete in order: a load issued
    // behind a store waits for it as well)
    uint64_t packed[2] = {0, 0};
    const uint32_t at = cnt;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (p < len && !bad) {
        const uint32_t bb = src[p];  // (a non-temporal load here: 9 % slower)
        if (bb > 32) {
          bad = true;
        } else {
          packed[i / 8] |= (uint64_t)bb << (8 * (i % 8));
          p += 1 + 16 * bb;
          ++cnt;
        }
      }
    }
    if (region != IDX_INVALID && at + 16 <= IDX_LOG) log[at / 16] = u64x2{packed[0], packed[1]};
  }
  slot_result[id] = bad ? IDX_INVALID : ((cnt << 16) | (p - len));
}

// A chunk's table entry with a pending slot resolved.
__device__ __forceinline__ uint32_t chunk_table_entry(const uint32_t *table, const uint32_t *slot_result, size_t c, uint32_t state) {
  uint32_t e = table[c * IDX_CAND + state];
  if (e != IDX_INVALID && (e & 0xFFFF) >= IDX_PENDING) {
    const uint32_t r = slot_result[c * IDX_SLOTS + (e & 0x7FFF)];
    e = r == IDX_INVALID ? IDX_INVALID : ((((e >> 16) + (r >> 16)) << 16) | (r & 0xFFFF));
  }
  return e;
}

constexpr int GROUP_THREADS = 576;  // one thread per candidate (513), whole waves

__global__ __launch_bounds__(GROUP_THREADS) void k_index_groups(const uint32_t *table, const uint32_t *slot_result, size_t nchunks,
                                                               uint64_t *gtable) {
  // the slot results of the group's chunks go to LDS first: the second of the two dependent loads per chunk then
  // costs an LDS read instead of a trip to memory
  __shared__ uint32_t slots[IDX_GROUP * IDX_SLOTS];
  const size_t cbeg = (size_t)blockIdx.x * IDX_GROUP;
  const size_t cend = cbeg + IDX_GROUP < nchunks ? cbeg + IDX_GROUP : nchunks;
  for (uint32_t i = threadIdx.x; i < (uint32_t)(cend - cbeg) * IDX_SLOTS; i += GROUP_THREADS) slots[i] = slot_result[cbeg * IDX_SLOTS + i];
  __syncthreads();
  if (threadIdx.x >= IDX_CAND) return;
  uint32_t state = threadIdx.x;
  uint64_t cnt = 0;
  bool bad = false;
  for (size_t c = cbeg; c < cend; ++c) {
    uint32_t e = table[c * IDX_CAND + state];
    if (e != IDX_INVALID && (e & 0xFFFF) >= IDX_PENDING) {
      const uint32_t r = slots[(c - cbeg) * IDX_SLOTS + (e & 0x7FFF)];
      e = r == IDX_INVALID ? IDX_INVALID : ((((e >> 16) + (r >> 16)) << 16) | (r & 0xFFFF));
    }
    if (e == IDX_INVALID) {
      bad = true;
      break;
    }
    cnt += e >> 16;
    state = e & 0xFFFF;
  }
  gtable[(size_t)blockIdx.x * IDX_CAND + threadIdx.x] = bad ? IDX_INVALID64 : ((cnt << 16) | state);
}

// Tables of IDX_SUPER consecutive groups composed into one (same walk as above over 64-bit entries).
__global__ __launch_bounds__(GROUP_THREADS) void k_index_supers(const uint64_t *gtable, size_t ngroups, uint64_t *stable) {
  if (threadIdx.x >= IDX_CAND) return;
  const size_t gbeg = (size_t)blockIdx.x * IDX_SUPER;
  const size_t gend = gbeg + IDX_SUPER < ngroups ? gbeg + IDX_SUPER : ngroups;
  uint64_t state = threadIdx.x, cnt = 0;
  bool bad = false;
  for (size_t g = gbeg; g < gend; ++g) {
    const uint64_t e = gtable[g * IDX_CAND + state];
    if (e == IDX_INVALID64) {
      bad = true;
      break;
    }
    cnt += e >> 16;
    state = e & 0xFFFF;
  }
  stable[(size_t)blockIdx.x * IDX_CAND + threadIdx.x] = bad ? IDX_INVALID64 : ((cnt << 16) | state);
}

// One thread per super-group: entry (first record number << 16 | entry byte) of each of its groups.
__global__ __launch_bounds__(64) void k_index_group_entries(const uint64_t *gtable, size_t ngroups, size_t nsupers,
                                                            const uint64_t *sentry, uint64_t *gentry) {
  const size_t sg = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (sg >= nsupers) return;
  const size_t gbeg = sg * IDX_SUPER;
  const size_t gend = gbeg + IDX_SUPER < ngroups ? gbeg + IDX_SUPER : ngroups;
  const uint64_t se = sentry[sg];
  bool bad = se == IDX_INVALID64;
  uint64_t state = se & 0xFFFF, base = se >> 16;
  for (size_t g = gbeg; g < gend; ++g) {
    if (bad) {
      gentry[g] = IDX_INVALID64;
      continue;
    }
    gentry[g] = (base << 16) | state;
    const uint64_t e = gtable[g * IDX_CAND + state];
    if (e == IDX_INVALID64) {
      bad = true;  // this group still hands entries to the chunks in front of the bad byte
      continue;
    }
    base += e >> 16;
    state = e & 0xFFFF;
  }
}

__global__ void k_index_spine(const uint64_t *gtable, size_t ngroups, size_t nbytes, uint64_t *gentry, uint64_t *result) {
  uint64_t state = 0, base = 0;
  bool bad = false;
  for (size_t g = 0; g < ngroups; ++g) {
    if (bad) {
      gentry[g] = IDX_INVALID64;
      continue;
    }
    gentry[g] = (base << 16) | state;
    const uint64_t e = gtable[g * IDX_CAND + state];
    if (e == IDX_INVALID64) {
      bad = true;
      continue;
    }
    base += e >> 16;
    state = e & 0xFFFF;
  }
  bad = bad || state != 0;  // the last record has to end on the last byte
  result[0] = base;
  result[1] = bad ? 0 : nbytes - base;
  result[2] = bad ? 1 : 0;
}

__global__ __launch_bounds__(64) void k_index_chunk_entries(const uint32_t *table, const uint32_t *slot_result, size_t nchunks, size_t ngroups,
                                                            const uint64_t *gentry, uint64_t *centry) {
  const size_t g = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (g >= ngroups) return;
  const size_t cbeg = g * IDX_GROUP;
  const size_t cend = cbeg + IDX_GROUP < nchunks ? cbeg + IDX_GROUP : nchunks;
  const uint64_t ge = gentry[g];
  bool bad = ge == IDX_INVALID64;
  uint64_t state = ge & 0xFFFF, base = ge >> 16;
  for (size_t c = cbeg; c < cend; ++c) {
    if (bad) {
      centry[c] = IDX_INVALID64;
      continue;
    }
    centry[c] = (base << 16) | state;
    const uint32_t e = chunk_table_entry(table, slot_result, c, state);
    if (e == IDX_INVALID) {
      bad = true;  // this chunk still emits the records in front of the bad byte
      continue;
    }
    base += e >> 16;
    state = e & 0xFFFF;
  }
}

// For the one-call decode: entries the emit pass will not write (a stream with fewer records than max_blocks; every
// entry when the stream is bad, the emit pass then writes those in front of the bad byte) read as zero-width blocks
// at offset 0.  Runs between the spine, which knows the record count, and the emit pass; usually finds nothing to do.
__global__ __launch_bounds__(256) void k_index_clear_tail(const uint64_t *result, size_t max_blocks, uint8_t *bits, uint64_t *off) {
  const size_t first = result[2] ? 0 : (result[0] < max_blocks ? result[0] : max_blocks);
  for (size_t i = first + (size_t)blockIdx.x * 256 + threadIdx.x; i < max_blocks; i += (size_t)gridDim.x * 256) {
    bits[i] = 0;
    off[i] = 0;
  }
}

constexpr int EMIT_THREADS = 64;
constexpr int EMIT_LANES = 16;  // lanes per chunk: one DPP row

// Kernel E: sixteen lanes per chunk write width and offset of every record of the chunk's true chain.  The chain is
// followed from global memory (all sixteen lanes read the same byte) until it arrives at the first byte of one of the
// chunk's slots; from there on the widths are in the slot's log, sixteen records per step: offsets by a prefix sum
// over the row.  A slot without a log, a log that is too short or a chain with a bad byte are walked to the end.
__global__ __launch_bounds__(EMIT_THREADS) void k_index_emit(const uint8_t *stream, size_t nbytes, const uint64_t *centry,
                                                             size_t nchunks, size_t max_blocks, const uint32_t *slot_start,
                                                             const uint32_t *slot_result, const uint32_t *slot_log,
                                                             const uint8_t *logs, uint8_t *bits, uint64_t *off) {
  const size_t c = ((size_t)blockIdx.x * EMIT_THREADS + threadIdx.x) / EMIT_LANES;
  const uint32_t sub = threadIdx.x % EMIT_LANES;
  const uint64_t ce = c < nchunks ? centry[c] : IDX_INVALID64;
  bool ok = ce != IDX_INVALID64;
  const size_t c0 = c * IDX_CHUNK;
  const uint8_t *src = stream + c0;
  const uint32_t len = !ok ? 0 : (uint32_t)(nbytes - c0 < (size_t)IDX_CHUNK ? nbytes - c0 : (size_t)IDX_CHUNK);
  uint32_t p = (uint32_t)(ce & 0xFFFF);
  uint64_t r = ce >> 16;
  uint32_t starts[IDX_SLOTS];
#pragma unroll
  for (int k = 0; k < IDX_SLOTS; ++k) starts[k] = ok ? slot_start[c * IDX_SLOTS + k] : IDX_INVALID;
  // (Staging the head of the chunk in LDS for these first steps, as the candidates kernel does, doubled the kernel's
  // time: 95 MB more to fetch for a dozen steps per chunk.)
  // (every lane of the row takes the same path through this function)  Records are handed to the lanes of the row
  // in turn and written sixteen at a time, so that the loads of the walk do not queue behind single stores.
  uint32_t held = 0, my_b = 0;
  size_t my_at = 0;
  auto flush = [&]() {
    if (sub < held && r - held + sub < max_blocks) {
      bits[r - held + sub] = (uint8_t)my_b;
      off[r - held + sub] = my_at - (r - held + sub);
    }
    held = 0;
  };
  auto walk = [&](bool to_slot) -> int {
    while (ok && p < len) {
      if (to_slot) {
#pragma unroll
        for (int k = 0; k < IDX_SLOTS; ++k)
          if (starts[k] == p) return k;
      }
      const uint32_t b = src[p];
      if (b > 32 || c0 + p + 1 + 16 * (size_t)b > nbytes) {
        // a bad width byte, or a record that runs past the end of the stream: the spine has already reported the
        // stream as bad; nothing is written for it, so that no later kernel reads a payload beyond the stream
        ok = false;
      } else {
        if (sub == held) {
          my_b = b;
          my_at = c0 + p;
        }
        p += 1 + 16 * b;
        ++r;
        if (++held == EMIT_LANES) flush();
      }
    }
    return -1;
  };
  const int slot = walk(true);
  flush();
  uint32_t logged = 0, records = 0;
  const uint8_t *log = logs;
  if (slot >= 0) {
    const uint32_t res = slot_result[c * IDX_SLOTS + slot], region = slot_log[c * IDX_SLOTS + slot];
    if (res != IDX_INVALID && region != IDX_INVALID) {
      records = res >> 16;
      logged = records < IDX_LOG ? records : IDX_LOG;
      log = logs + (size_t)region * IDX_LOG;
    }
  }
  for (uint32_t i0 = 0; __any(i0 < logged); i0 += EMIT_LANES) {  // (the rows of a wave stay in step for the DPP scan)
    const uint32_t i = i0 + sub;
    const bool have = i < logged;
    const uint32_t b = have ? log[i] : 0;
    const uint32_t size = have ? 1 + 16 * b : 0;
    uint32_t incl = size;
    incl += dpp0<DPP_ROW_SHR1>(incl);
    incl += dpp0<DPP_ROW_SHR2>(incl);
    incl += dpp0<DPP_ROW_SHR4>(incl);
    incl += dpp0<DPP_ROW_SHR8>(incl);
    const size_t at = c0 + p + incl - size;
    if (have && at + size <= nbytes && r + i0 + sub < max_blocks) {
      bits[r + i] = (uint8_t)b;
      off[r + i] = at - (r + i);
    }
    p += __shfl(incl, (int)((threadIdx.x & ~(EMIT_LANES - 1)) + EMIT_LANES - 1));
  }
  r += logged;
  if (slot >= 0 && logged == records && records != 0) return;  // the log held the rest of the chunk
  walk(false);
  flush();
}

__global__ void k_stream_bytes(const uint64_t *payload_bytes, uint64_t nblocks, uint64_t *out) { *out = *payload_bytes + nblocks; }

hipError_t launch_stream_bytes(const uint64_t *payload_bytes, size_t nblocks, uint64_t *out, hipStream_t st) {
  hipLaunchKernelGGL(k_stream_bytes, dim3(1), dim3(1), 0, st, payload_bytes, (uint64_t)nblocks, out);
  return hipGetLastError();
}

static size_t idx_chunks(size_t nbytes) { return (nbytes + IDX_CHUNK - 1) / IDX_CHUNK; }

static size_t idx_groups(size_t nchunks) { return (nchunks + IDX_GROUP - 1) / IDX_GROUP; }
static size_t idx_supers(size_t ngroups) { return (ngroups + IDX_SUPER - 1) / IDX_SUPER; }

size_t stream_index_workspace_bytes(size_t nbytes) {
  const size_t nchunks = idx_chunks(nbytes), ngroups = idx_groups(nchunks), nsupers = idx_supers(ngroups);
  return align256(4 * IDX_CAND * nchunks) + align256(8 * IDX_CAND * ngroups) + align256(8 * IDX_CAND * nsupers) +
         align256(8 * nsupers) + align256(8 * ngroups) + align256(8 * nchunks) + 3 * align256(4 * IDX_SLOTS * nchunks) + 256 +
         align256(idx_log_regions(nchunks) * IDX_LOG) + 256;
}

hipError_t launch_stream_index(const uint8_t *stream, size_t nbytes, size_t max_blocks, uint8_t *bits, uint64_t *off,
                               uint64_t *result, void *workspace, bool clear_tail, hipStream_t st) {
  if (nbytes == 0) {
    if (clear_tail && max_blocks) {
      if (hipError_t e = hipMemsetAsync(bits, 0, max_blocks, st)) return e;
      if (hipError_t e = hipMemsetAsync(off, 0, 8 * max_blocks, st)) return e;
    }
    return hipMemsetAsync(result, 0, 3 * sizeof(uint64_t), st);
  }
  const size_t nchunks = idx_chunks(nbytes), ngroups = idx_groups(nchunks), nsupers = idx_supers(ngroups);
  uint8_t *ws = static_cast<uint8_t *>(workspace);
  uint32_t *table = reinterpret_cast<uint32_t *>(ws);
  ws += align256(4 * IDX_CAND * nchunks);
  uint64_t *gtable = reinterpret_cast<uint64_t *>(ws);
  ws += align256(8 * IDX_CAND * ngroups);
  uint64_t *stable = reinterpret_cast<uint64_t *>(ws);
  ws += align256(8 * IDX_CAND * nsupers);
  uint64_t *sentry = reinterpret_cast<uint64_t *>(ws);
  ws += align256(8 * nsupers);
  uint64_t *gentry = reinterpret_cast<uint64_t *>(ws);
  ws += align256(8 * ngroups);
  uint64_t *centry = reinterpret_cast<uint64_t *>(ws);
  ws += align256(8 * nchunks);
  uint32_t *slot_start = reinterpret_cast<uint32_t *>(ws);
  ws += align256(4 * IDX_SLOTS * nchunks);
  uint32_t *slot_result = reinterpret_cast<uint32_t *>(ws);
  ws += align256(4 * IDX_SLOTS * nchunks);
  uint32_t *slot_log = reinterpret_cast<uint32_t *>(ws);
  ws += align256(4 * IDX_SLOTS * nchunks);
  uint32_t *pool_next = reinterpret_cast<uint32_t *>(ws);
  ws += 256;
  uint8_t *logs = ws;
  const size_t nslots = nchunks * IDX_SLOTS, nregions = idx_log_regions(nchunks);
  if (nregions > 0xFFFFFFF0ull) return hipErrorInvalidValue;  // (a stream of more than 40 TiB)
  hipLaunchKernelGGL(k_index_candidates, dim3((unsigned)nchunks), dim3(CAND_THREADS), 0, st, stream, nbytes, table, slot_start,
                     pool_next);
  hipLaunchKernelGGL(k_index_survivors, dim3((unsigned)((nslots + SURV_THREADS - 1) / SURV_THREADS)), dim3(SURV_THREADS), 0, st,
                     stream, nbytes, nslots, slot_start, slot_result, slot_log, pool_next, (uint32_t)nregions, logs);
  hipLaunchKernelGGL(k_index_groups, dim3((unsigned)ngroups), dim3(GROUP_THREADS), 0, st, table, slot_result, nchunks, gtable);
  hipLaunchKernelGGL(k_index_supers, dim3((unsigned)nsupers), dim3(GROUP_THREADS), 0, st, gtable, ngroups, stable);
  hipLaunchKernelGGL(k_index_spine, dim3(1), dim3(1), 0, st, stable, nsupers, nbytes, sentry, result);
  if (clear_tail && max_blocks) hipLaunchKernelGGL(k_index_clear_tail, dim3(512), dim3(256), 0, st, result, max_blocks, bits, off);
  hipLaunchKernelGGL(k_index_group_entries, dim3((unsigned)((nsupers + 63) / 64)), dim3(64), 0, st, gtable, ngroups, nsupers,
                     sentry, gentry);
  hipLaunchKernelGGL(k_index_chunk_entries, dim3((unsigned)((ngroups + 63) / 64)), dim3(64), 0, st, table, slot_result, nchunks, ngroups,
                     gentry, centry);
  constexpr size_t per_wg = EMIT_THREADS / EMIT_LANES;
  hipLaunchKernelGGL(k_index_emit, dim3((unsigned)((nchunks + per_wg - 1) / per_wg)), dim3(EMIT_THREADS), 0, st, stream, nbytes, centry,
                     nchunks, max_blocks, slot_start, slot_result, slot_log, logs, bits, off);
  return hipGetLastError();
}

}  // namespace sc
