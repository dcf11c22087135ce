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

namespace sc {

// ---------------------------------------------------------------------------
// chained seeds
// ---------------------------------------------------------------------------
// sums[k] = sum of the 128 b-bit fields of block k (its deltas), or, for a
// 32-bit block, its last raw value.  One block per half-wave, same extraction
// as the decoder (wave-private LDS slot, funnel shift), nothing is stored but
// one word per block.
__global__ __launch_bounds__(256) void k_block_sums(const uint8_t *arena, const uint64_t *off, const uint8_t *bits,
                                                    size_t nblocks, uint32_t *sums) {
  __shared__ u32x4 lds[4][2][33];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  u32x4 *slot = lds[wv][h];
  const size_t blk = ((size_t)blockIdx.x * 4 + wv) * 2 + h;
  const bool valid = blk < nblocks;
  const size_t bc = valid ? blk : nblocks - 1;
  uint32_t b = bits[bc];
  b = b > 32 ? 32 : b;
  const u32x4 *src = reinterpret_cast<const u32x4 *>(arena + off[bc]);
  u32x4 ld = {0, 0, 0, 0};
  if ((uint32_t)k < b) ld = src[k];
  wave_lds_fence();
  slot[k] = ld;
  wave_lds_fence();
  const uint32_t p = (uint32_t)k * b, w = p >> 5, s = p & 31;
  const u32x4 lo = slot[w], hi = slot[w + 1];
  const uint32_t m = field_mask(b);
  u32x4 v;
  v.x = funnel(hi.x, lo.x, s) & m;
  v.y = funnel(hi.y, lo.y, s) & m;
  v.z = funnel(hi.z, lo.z, s) & m;
  v.w = funnel(hi.w, lo.w, s) & m;
  const uint32_t total = half_scan_add(v.x + v.y + v.z + v.w);  // lane 31 of the half-wave: the block's sum
  if (valid && k == 31) sums[blk] = b == 32 ? v.w : total;
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

__global__ __launch_bounds__(256) void k_chain_tiles(const uint8_t *bits, const uint32_t *sums, size_t n, uint64_t *tile_sum) {
  __shared__ uint64_t sh[256];
  const size_t base = (size_t)blockIdx.x * CHAIN_TILE + (size_t)threadIdx.x * 16;
  uint64_t e = 0;
  for (int i = 0; i < 16; ++i)
    if (base + i < n) e = chain_compose(e, chain_element(bits[base + i], sums[base + i]));
  chain_wg_scan(e, sh);
  if (threadIdx.x == 255) tile_sum[blockIdx.x] = sh[255];
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
  __shared__ uint64_t sh[256];
  const size_t base = (size_t)blockIdx.x * CHAIN_TILE + (size_t)threadIdx.x * 16;
  uint64_t el[16];
  uint64_t e = 0;
  for (int i = 0; i < 16; ++i) {
    el[i] = base + i < n ? chain_element(bits[base + i], sums[base + i]) : 0;
    e = chain_compose(e, el[i]);
  }
  chain_wg_scan(e, sh);
  uint64_t state = chain_compose((1ull << 32) | tile_in[blockIdx.x], threadIdx.x ? sh[threadIdx.x - 1] : 0);
  for (int i = 0; i < 16; ++i) {
    if (base + i < n) seeds[base + i] = (uint32_t)state;
    state = chain_compose(state, el[i]);
  }
}

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

size_t chain_workspace_bytes(size_t nblocks) {
  const size_t ntiles = (nblocks + CHAIN_TILE - 1) / CHAIN_TILE;
  return align256(4 * nblocks) + align256(8 * ntiles) + align256(4 * ntiles) + 256;
}

hipError_t launch_chain_offsets(uint32_t init, const void *arena, const uint64_t *off, const uint8_t *bits, size_t nblocks,
                                uint32_t *seeds_out, void *workspace, hipStream_t st) {
  if (nblocks == 0) return hipSuccess;
  const size_t ntiles = (nblocks + CHAIN_TILE - 1) / CHAIN_TILE;
  uint8_t *ws = static_cast<uint8_t *>(workspace);
  uint32_t *sums = reinterpret_cast<uint32_t *>(ws);
  uint64_t *tile_sum = reinterpret_cast<uint64_t *>(ws + align256(4 * nblocks));
  uint32_t *tile_in = reinterpret_cast<uint32_t *>(ws + align256(4 * nblocks) + align256(8 * ntiles));
  hipLaunchKernelGGL(k_block_sums, dim3((unsigned)((nblocks + 7) / 8)), dim3(256), 0, st, static_cast<const uint8_t *>(arena), off,
                     bits, nblocks, sums);
  hipLaunchKernelGGL(k_chain_tiles, dim3((unsigned)ntiles), dim3(256), 0, st, bits, sums, nblocks, tile_sum);
  hipLaunchKernelGGL(k_chain_spine, dim3(1), dim3(256), 0, st, tile_sum, ntiles, init, tile_in);
  hipLaunchKernelGGL(k_chain_apply, dim3((unsigned)ntiles), dim3(256), 0, st, bits, sums, nblocks, tile_in, seeds_out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// [b][payload] stream <-> aligned arena
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t align_bytes(uint32_t hi, uint32_t lo, uint32_t nbytes) {
  return __builtin_amdgcn_alignbyte(hi, lo, nbytes);  // ({hi,lo} >> 8*nbytes)[31:0]
}

// One block per half-wave; thread k < b moves payload vector k.  The destination is at an arbitrary byte alignment:
// bytes up to the next 4-byte boundary and after the last one go out as byte stores, the rest as aligned words.
__global__ __launch_bounds__(256) void k_stream_build(const uint8_t *arena, const uint64_t *off, const uint8_t *bits, size_t nblocks,
                                                      uint8_t *stream) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  const size_t blk = ((size_t)blockIdx.x * 4 + wv) * 2 + h;
  if (blk >= nblocks) return;
  uint32_t b = bits[blk];
  b = b > 32 ? 32 : b;
  const uint64_t o = off[blk];
  uint8_t *rec = stream + o + blk;
  if (k == 0) rec[0] = (uint8_t)b;
  if ((uint32_t)k >= b) return;
  const u32x4 x = reinterpret_cast<const u32x4 *>(arena + o)[k];
  uint8_t *d = rec + 1 + 16 * k;
  const uint32_t m = (uint32_t)((uintptr_t)d & 3);
  if (m == 0) {
    uint32_t *q = reinterpret_cast<uint32_t *>(d);
    q[0] = x.x;
    q[1] = x.y;
    q[2] = x.z;
    q[3] = x.w;
    return;
  }
  const uint32_t head = 4 - m;  // bytes before the first aligned word
  for (uint32_t i = 0; i < head; ++i) d[i] = (uint8_t)(x.x >> (8 * i));
  uint32_t *q = reinterpret_cast<uint32_t *>(d + head);
  q[0] = align_bytes(x.y, x.x, head);
  q[1] = align_bytes(x.z, x.y, head);
  q[2] = align_bytes(x.w, x.z, head);
  for (uint32_t i = 0; i < m; ++i) d[head + 12 + i] = (uint8_t)(x.w >> (8 * (head + i)));
}

// The inverse: five aligned words cover a payload vector that starts m bytes into the first of them.
__global__ __launch_bounds__(256) void k_stream_unbuild(const uint8_t *stream, const uint64_t *off, const uint8_t *bits,
                                                        size_t nblocks, uint8_t *arena) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int h = lane >> 5, k = lane & 31;
  const size_t blk = ((size_t)blockIdx.x * 4 + wv) * 2 + h;
  if (blk >= nblocks) return;
  uint32_t b = bits[blk];
  b = b > 32 ? 32 : b;
  if ((uint32_t)k >= b) return;
  const uint64_t o = off[blk];
  const uint8_t *s = stream + o + blk + 1 + 16 * k;
  const uint32_t m = (uint32_t)((uintptr_t)s & 3);
  const uint32_t *q = reinterpret_cast<const uint32_t *>(s - m);
  const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
  const uint32_t d4 = m ? q[4] : 0;  // the word past the vector is only touched when it holds bytes of it
  u32x4 x;
  x.x = align_bytes(d1, d0, m);
  x.y = align_bytes(d2, d1, m);
  x.z = align_bytes(d3, d2, m);
  x.w = align_bytes(d4, d3, m);
  reinterpret_cast<u32x4 *>(arena + o)[k] = x;
}

hipError_t launch_stream_build(const void *arena, const uint64_t *off, const uint8_t *bits, size_t nblocks, uint8_t *stream,
                               hipStream_t st) {
  if (nblocks == 0) return hipSuccess;
  hipLaunchKernelGGL(k_stream_build, dim3((unsigned)((nblocks + 7) / 8)), dim3(256), 0, st, static_cast<const uint8_t *>(arena), off,
                     bits, nblocks, stream);
  return hipGetLastError();
}

hipError_t launch_stream_unbuild(const uint8_t *stream, const uint64_t *off, const uint8_t *bits, size_t nblocks, void *arena,
                                 hipStream_t st) {
  if (nblocks == 0) return hipSuccess;
  hipLaunchKernelGGL(k_stream_unbuild, dim3((unsigned)((nblocks + 7) / 8)), dim3(256), 0, st, stream, off, bits, nblocks,
                     static_cast<uint8_t *>(arena));
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// stream index: where do the records start?
// ---------------------------------------------------------------------------
// Record k+1 starts 1 + 16*b_k bytes after record k: a chain through the data.  It is cut into chunks of IDX_CHUNK
// bytes.  The first record of a chunk starts at one of 513 places (a record is at most 513 bytes long), so
//   A  every chunk walks the chain from all 513 candidate entries at once (chunk staged in LDS) and tabulates, per
//      candidate, where the walk leaves the chunk (0..512 bytes into the next one) and how many records it saw;
//   B  groups of IDX_GROUP chunks compose their tables (thread f follows candidate f through the group);
//   C  one thread walks the group tables from the known entry of the stream (byte 0) and fixes every group's entry
//      and first record number;  D  one thread per group does the same for its chunks;
//   E  every chunk walks its true chain once more and writes width and arena offset of each record.
// A width byte above 32 ends a walk as invalid; the true chain hitting one is reported in result[2].
constexpr int IDX_CHUNK = 32768;
constexpr int IDX_CAND = 513;
constexpr int IDX_THREADS = 576;
constexpr int IDX_GROUP = 256;
constexpr uint32_t IDX_INVALID = 0xFFFFFFFFu;
constexpr uint64_t IDX_INVALID64 = ~0ull;
constexpr int IDX_BATCH = 2048;

// Stage chunk c (len bytes from stream + c * IDX_CHUNK) in LDS as aligned words; returns the byte view of its start.
template <int THREADS>
__device__ __forceinline__ const uint8_t *stage_chunk(const uint8_t *stream, size_t c0, uint32_t len, uint32_t *sh) {
  const uintptr_t a0 = (uintptr_t)(stream + c0);
  const uint32_t m = (uint32_t)(a0 & 3);
  const uint32_t *q = reinterpret_cast<const uint32_t *>(a0 - m);
  const uint32_t nwords = (m + len + 3) / 4;
  for (uint32_t i = threadIdx.x; i < nwords; i += THREADS) sh[i] = q[i];
  __syncthreads();
  return reinterpret_cast<const uint8_t *>(sh) + m;
}

__global__ __launch_bounds__(IDX_THREADS) void k_index_candidates(const uint8_t *stream, size_t nbytes, uint32_t *table) {
  __shared__ uint32_t sh[IDX_CHUNK / 4 + 1];
  const size_t c0 = (size_t)blockIdx.x * IDX_CHUNK;
  const uint32_t len = (uint32_t)(nbytes - c0 < (size_t)IDX_CHUNK ? nbytes - c0 : (size_t)IDX_CHUNK);
  const uint8_t *by = stage_chunk<IDX_THREADS>(stream, c0, len, sh);
  if (threadIdx.x >= IDX_CAND) return;
  uint32_t p = threadIdx.x, cnt = 0;
  bool bad = false;
  while (p < len) {
    const uint32_t b = by[p];
    if (b > 32) {
      bad = true;
      break;
    }
    p += 1 + 16 * b;
    ++cnt;
  }
  table[(size_t)blockIdx.x * IDX_CAND + threadIdx.x] = bad ? IDX_INVALID : ((cnt << 16) | (p - len));
}

__global__ __launch_bounds__(IDX_THREADS) void k_index_groups(const uint32_t *table, size_t nchunks, uint64_t *gtable) {
  if (threadIdx.x >= IDX_CAND) return;
  const size_t cbeg = (size_t)blockIdx.x * IDX_GROUP;
  const size_t cend = cbeg + IDX_GROUP < nchunks ? cbeg + IDX_GROUP : nchunks;
  uint32_t state = threadIdx.x;
  uint64_t cnt = 0;
  bool bad = false;
  for (size_t c = cbeg; c < cend; ++c) {
    const uint32_t e = table[c * IDX_CAND + state];
    if (e == IDX_INVALID) {
      bad = true;
      break;
    }
    cnt += e >> 16;
    state = e & 0xFFFF;
  }
  gtable[(size_t)blockIdx.x * IDX_CAND + threadIdx.x] = bad ? IDX_INVALID64 : ((cnt << 16) | state);
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

__global__ __launch_bounds__(64) void k_index_chunk_entries(const uint32_t *table, size_t nchunks, size_t ngroups,
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
    const uint32_t e = table[c * IDX_CAND + state];
    if (e == IDX_INVALID) {
      bad = true;  // this chunk still emits the records in front of the bad byte
      continue;
    }
    base += e >> 16;
    state = e & 0xFFFF;
  }
}

__global__ __launch_bounds__(64) void k_index_emit(const uint8_t *stream, size_t nbytes, const uint64_t *centry, size_t max_blocks,
                                                   uint8_t *bits, uint64_t *off) {
  __shared__ uint32_t sh[IDX_CHUNK / 4 + 1];
  __shared__ uint16_t plist[IDX_BATCH];
  __shared__ uint32_t s_p, s_n;
  const uint64_t ce = centry[blockIdx.x];
  if (ce == IDX_INVALID64) return;
  const size_t c0 = (size_t)blockIdx.x * IDX_CHUNK;
  const uint32_t len = (uint32_t)(nbytes - c0 < (size_t)IDX_CHUNK ? nbytes - c0 : (size_t)IDX_CHUNK);
  const uint8_t *by = stage_chunk<64>(stream, c0, len, sh);
  uint64_t base = ce >> 16;
  if (threadIdx.x == 0) s_p = (uint32_t)(ce & 0xFFFF);
  __syncthreads();
  for (;;) {
    if (threadIdx.x == 0) {
      uint32_t p = s_p, n = 0;
      while (p < len && n < IDX_BATCH) {
        const uint32_t b = by[p];
        if (b > 32) {
          p = len;  // stop here: the spine has already reported the stream as bad
          break;
        }
        plist[n++] = (uint16_t)p;
        p += 1 + 16 * b;
      }
      s_p = p;
      s_n = n;
    }
    __syncthreads();
    const uint32_t n = s_n;
    for (uint32_t i = threadIdx.x; i < n; i += 64) {
      const uint64_t r = base + i;
      if (r < max_blocks) {
        const uint32_t p = plist[i];
        bits[r] = by[p];
        off[r] = (uint64_t)c0 + p - r;
      }
    }
    base += n;
    const bool more = s_p < len && n == IDX_BATCH;
    __syncthreads();
    if (!more) break;
  }
}

static size_t idx_chunks(size_t nbytes) { return (nbytes + IDX_CHUNK - 1) / IDX_CHUNK; }

size_t stream_index_workspace_bytes(size_t nbytes) {
  const size_t nchunks = idx_chunks(nbytes), ngroups = (nchunks + IDX_GROUP - 1) / IDX_GROUP;
  return align256(4 * IDX_CAND * nchunks) + align256(8 * IDX_CAND * ngroups) + align256(8 * ngroups) + align256(8 * nchunks) + 256;
}

hipError_t launch_stream_index(const uint8_t *stream, size_t nbytes, size_t max_blocks, uint8_t *bits, uint64_t *off,
                               uint64_t *result, void *workspace, hipStream_t st) {
  if (nbytes == 0) return hipMemsetAsync(result, 0, 3 * sizeof(uint64_t), st);
  const size_t nchunks = idx_chunks(nbytes), ngroups = (nchunks + IDX_GROUP - 1) / IDX_GROUP;
  uint8_t *ws = static_cast<uint8_t *>(workspace);
  uint32_t *table = reinterpret_cast<uint32_t *>(ws);
  ws += align256(4 * IDX_CAND * nchunks);
  uint64_t *gtable = reinterpret_cast<uint64_t *>(ws);
  ws += align256(8 * IDX_CAND * ngroups);
  uint64_t *gentry = reinterpret_cast<uint64_t *>(ws);
  ws += align256(8 * ngroups);
  uint64_t *centry = reinterpret_cast<uint64_t *>(ws);
  hipLaunchKernelGGL(k_index_candidates, dim3((unsigned)nchunks), dim3(IDX_THREADS), 0, st, stream, nbytes, table);
  hipLaunchKernelGGL(k_index_groups, dim3((unsigned)ngroups), dim3(IDX_THREADS), 0, st, table, nchunks, gtable);
  hipLaunchKernelGGL(k_index_spine, dim3(1), dim3(1), 0, st, gtable, ngroups, nbytes, gentry, result);
  hipLaunchKernelGGL(k_index_chunk_entries, dim3((unsigned)((ngroups + 63) / 64)), dim3(64), 0, st, table, nchunks, ngroups, gentry,
                     centry);
  hipLaunchKernelGGL(k_index_emit, dim3((unsigned)nchunks), dim3(64), 0, st, stream, nbytes, centry, max_blocks, bits, off);
  return hipGetLastError();
}

}  // namespace sc
