// extern "C" batched device-pointer entry points declared in include/simdcomp_hip.h:
// argument checks, then a kernel launch on the caller's stream.  Nothing here
// allocates, frees or synchronises.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "../../include/simdcomp_hip.h"
#include "codec_kernels.h"
#include "stream_kernels.h"
#include "query_kernels.h"
#include "api_error.h"

namespace {

int fail(int code, const char *what) { return sc::api_fail(code, what); }

int hip_result(hipError_t e, const char *where) {
  if (e == hipSuccess) return SIMDCOMP_HIP_OK;
  char msg[256];
  snprintf(msg, sizeof msg, "%s: %s", where, hipGetErrorString(e));
  return sc::api_fail((int)e, msg);
}

// The launchers report hipGetLastError() behind their launches, and HIP keeps a thread's last error until somebody
// reads it: an earlier failure of the application's own (a hipHostRegister that was refused, say) would come back as
// this call's result.  So the error state is read, i.e. reset, right before the launch.
#define SC_LAUNCH(call, what) ((void)hipGetLastError(), hip_result((call), (what)))

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

int check_common(const void *ints, const void *payload, size_t nblocks, uint32_t bit, bool need_payload) {
  if (bit > 32) return fail(SIMDCOMP_HIP_EBIT, "bit width > 32");
  if (nblocks == 0) return SIMDCOMP_HIP_OK;
  if (!ints) return fail(SIMDCOMP_HIP_ENULL, "integer array is NULL");
  if (need_payload && !payload) return fail(SIMDCOMP_HIP_ENULL, "payload pointer is NULL");
  if (!aligned16(ints) || !aligned16(payload)) return fail(SIMDCOMP_HIP_EALIGN, "pointer not 16-byte aligned");
  return SIMDCOMP_HIP_OK;
}

int do_pack(int mode, int masked, uint32_t initvalue, const uint32_t *seeds, const uint32_t *in, void *payload,
            const uint64_t *off, const uint8_t *bits, uint32_t bit, size_t nblocks, uint32_t *seeds_out,
            void *stream, bool stream_layout = false) {
  const bool mixed = bits != nullptr;
  // the byte stream has no alignment of its own: only the integer side is checked then
  if (int rc = check_common(in, stream_layout ? in : payload, nblocks, mixed ? 0 : bit, mixed || bit != 0)) return rc;
  if (stream_layout && nblocks && (!payload || !mixed)) return fail(SIMDCOMP_HIP_ENULL, "stream layout needs a stream and per-block widths");
  if (mixed && !off && nblocks) return fail(SIMDCOMP_HIP_ENULL, "per-block widths need per-block payload offsets");
  if (mode == SIMDCOMP_HIP_FOR && !seeds && mixed && nblocks)
    return fail(SIMDCOMP_HIP_ENULL, "frame bases are NULL");
  sc::PackArgs a{};
  a.in = in;
  a.payload = payload;
  a.off = off;
  a.bits = bits;
  a.seeds = seeds;
  a.seeds_out = seeds_out;
  a.nblocks = nblocks;
  a.length = nblocks * 128;
  a.bit = bit;
  a.seed0 = initvalue;
  a.carry_blocks = 0;
  a.stream_layout = stream_layout ? 1u : 0u;
  return SC_LAUNCH(sc::launch_pack(mode, masked != 0 && mode != SIMDCOMP_HIP_FOR, false, a,
                                    as_stream(stream)),
                    "pack launch");
}

int do_unpack(int mode, uint32_t seed0, const uint32_t *seeds, const void *payload, const uint64_t *off,
              const uint8_t *bits, uint32_t bit, uint32_t *out, size_t nblocks, void *stream, bool stream_layout = false) {
  const bool mixed = bits != nullptr;
  if (int rc = check_common(out, stream_layout ? out : payload, nblocks, mixed ? 0 : bit, mixed || bit != 0)) return rc;
  if (stream_layout && nblocks && (!payload || !mixed)) return fail(SIMDCOMP_HIP_ENULL, "stream layout needs a stream and per-block widths");
  if (mixed && !off && nblocks) return fail(SIMDCOMP_HIP_ENULL, "per-block widths need per-block payload offsets");
  sc::UnpackArgs a{};
  a.payload = payload;
  a.off = off;
  a.bits = bits;
  a.seeds = seeds;
  a.out = out;
  a.nblocks = nblocks;
  a.length = nblocks * 128;
  a.bit = bit;
  a.seed0 = seed0;
  a.stream_layout = stream_layout ? 1u : 0u;
  return SC_LAUNCH(sc::launch_unpack(mode, false, a, as_stream(stream)), "unpack launch");
}

}  // namespace

extern "C" {

int simdcomp_hip_pack(const uint32_t *d_in, void *d_out, size_t nblocks, uint32_t bit, int masked, void *stream) {
  return do_pack(sc::MODE_PLAIN, masked, 0, nullptr, d_in, d_out, nullptr, nullptr, bit, nblocks, nullptr, stream);
}

int simdcomp_hip_unpack(const void *d_in, uint32_t *d_out, size_t nblocks, uint32_t bit, void *stream) {
  return do_unpack(sc::MODE_PLAIN, 0, nullptr, d_in, nullptr, nullptr, bit, d_out, nblocks, stream);
}

int simdcomp_hip_packd1(uint32_t initvalue, const uint32_t *d_initvalues, const uint32_t *d_in, void *d_out,
                        size_t nblocks, uint32_t bit, int masked, uint32_t *d_initvalues_out, void *stream) {
  return do_pack(sc::MODE_D1, masked, initvalue, d_initvalues, d_in, d_out, nullptr, nullptr, bit, nblocks,
                 d_initvalues_out, stream);
}

int simdcomp_hip_unpackd1(const uint32_t *d_initvalues, const void *d_in, uint32_t *d_out, size_t nblocks,
                          uint32_t bit, void *stream) {
  if (!d_initvalues && nblocks) return fail(SIMDCOMP_HIP_ENULL, "d_initvalues is NULL");
  return do_unpack(sc::MODE_D1, 0, d_initvalues, d_in, nullptr, nullptr, bit, d_out, nblocks, stream);
}

int simdcomp_hip_packFOR(const uint32_t *d_initvalues, const uint32_t *d_in, void *d_out, size_t nblocks,
                         uint32_t bit, void *stream) {
  if (!d_initvalues && nblocks) return fail(SIMDCOMP_HIP_ENULL, "d_initvalues is NULL");
  return do_pack(sc::MODE_FOR, 0, 0, d_initvalues, d_in, d_out, nullptr, nullptr, bit, nblocks, nullptr, stream);
}

int simdcomp_hip_unpackFOR(const uint32_t *d_initvalues, const void *d_in, uint32_t *d_out, size_t nblocks,
                           uint32_t bit, void *stream) {
  if (!d_initvalues && nblocks) return fail(SIMDCOMP_HIP_ENULL, "d_initvalues is NULL");
  return do_unpack(sc::MODE_FOR, 0, d_initvalues, d_in, nullptr, nullptr, bit, d_out, nblocks, stream);
}

int simdcomp_hip_pack_mixed(int mode, int masked, uint32_t initvalue, const uint32_t *d_initvalues,
                            const uint32_t *d_in, void *d_payload, const uint64_t *d_payload_offsets,
                            const uint8_t *d_bits, size_t nblocks, uint32_t *d_initvalues_out, void *stream) {
  if (mode < 0 || mode > 2) return fail(SIMDCOMP_HIP_EMODE, "unknown mode");
  if (!d_bits && nblocks) return fail(SIMDCOMP_HIP_ENULL, "d_bits is NULL");
  return do_pack(mode, masked, initvalue, d_initvalues, d_in, d_payload, d_payload_offsets, d_bits, 0, nblocks,
                 d_initvalues_out, stream);
}

int simdcomp_hip_unpack_mixed(int mode, const uint32_t *d_initvalues, const void *d_payload,
                              const uint64_t *d_payload_offsets, const uint8_t *d_bits, uint32_t *d_out,
                              size_t nblocks, void *stream) {
  if (mode < 0 || mode > 2) return fail(SIMDCOMP_HIP_EMODE, "unknown mode");
  if (!d_bits && nblocks) return fail(SIMDCOMP_HIP_ENULL, "d_bits is NULL");
  if (mode != SIMDCOMP_HIP_PLAIN && !d_initvalues && nblocks) return fail(SIMDCOMP_HIP_ENULL, "d_initvalues is NULL");
  return do_unpack(mode, 0, d_initvalues, d_payload, d_payload_offsets, d_bits, 0, d_out, nblocks, stream);
}

int simdcomp_hip_pack_stream(int mode, int masked, uint32_t initvalue, const uint32_t *d_initvalues,
                             const uint32_t *d_in, void *d_stream, const uint64_t *d_payload_offsets,
                             const uint8_t *d_bits, size_t nblocks, uint32_t *d_initvalues_out, void *stream) {
  if (mode < 0 || mode > 2) return fail(SIMDCOMP_HIP_EMODE, "unknown mode");
  if (nblocks && (!d_bits || !d_payload_offsets)) return fail(SIMDCOMP_HIP_ENULL, "widths / offsets are NULL");
  return do_pack(mode, masked, initvalue, d_initvalues, d_in, d_stream, d_payload_offsets, d_bits, 0, nblocks,
                 d_initvalues_out, stream, true);
}

int simdcomp_hip_unpack_stream(int mode, const uint32_t *d_initvalues, const void *d_stream,
                               const uint64_t *d_payload_offsets, const uint8_t *d_bits, uint32_t *d_out,
                               size_t nblocks, void *stream) {
  if (mode < 0 || mode > 2) return fail(SIMDCOMP_HIP_EMODE, "unknown mode");
  if (nblocks && (!d_bits || !d_payload_offsets)) return fail(SIMDCOMP_HIP_ENULL, "widths / offsets are NULL");
  if (mode != SIMDCOMP_HIP_PLAIN && !d_initvalues && nblocks) return fail(SIMDCOMP_HIP_ENULL, "d_initvalues is NULL");
  return do_unpack(mode, 0, d_initvalues, d_stream, d_payload_offsets, d_bits, 0, d_out, nblocks, stream, true);
}

int simdcomp_hip_maxbits(const uint32_t *d_in, size_t nblocks, uint8_t *d_bits, void *stream) {
  if (nblocks == 0) return SIMDCOMP_HIP_OK;
  if (!d_in || !d_bits) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (!aligned16(d_in)) return fail(SIMDCOMP_HIP_EALIGN, "pointer not 16-byte aligned");
  sc::ReduceArgs a{};
  a.in = d_in;
  a.bits_out = d_bits;
  a.nblocks = nblocks;
  return SC_LAUNCH(sc::launch_reduce(sc::RED_MAXBITS, a, as_stream(stream)), "maxbits launch");
}

int simdcomp_hip_maxbitsd1(uint32_t initvalue, const uint32_t *d_initvalues, const uint32_t *d_in, size_t nblocks,
                           uint8_t *d_bits, uint32_t *d_initvalues_out, void *stream) {
  if (nblocks == 0) return SIMDCOMP_HIP_OK;
  if (!d_in || !d_bits) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (!aligned16(d_in)) return fail(SIMDCOMP_HIP_EALIGN, "pointer not 16-byte aligned");
  sc::ReduceArgs a{};
  a.in = d_in;
  a.seeds = d_initvalues;
  a.seeds_out = d_initvalues_out;
  a.bits_out = d_bits;
  a.nblocks = nblocks;
  a.seed0 = initvalue;
  return SC_LAUNCH(sc::launch_reduce(sc::RED_MAXBITSD1, a, as_stream(stream)), "maxbitsd1 launch");
}

int simdcomp_hip_maxmin(const uint32_t *d_in, size_t nblocks, uint32_t *d_min, uint32_t *d_max, uint8_t *d_bits,
                        void *stream) {
  if (nblocks == 0) return SIMDCOMP_HIP_OK;
  if (!d_in) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (!aligned16(d_in)) return fail(SIMDCOMP_HIP_EALIGN, "pointer not 16-byte aligned");
  sc::ReduceArgs a{};
  a.in = d_in;
  a.min_out = d_min;
  a.max_out = d_max;
  a.bits_out = d_bits;
  a.nblocks = nblocks;
  return SC_LAUNCH(sc::launch_reduce(sc::RED_MINMAX, a, as_stream(stream)), "maxmin launch");
}

int simdcomp_hip_reduce_length(const uint32_t *d_in, size_t length, uint32_t initvalue, uint32_t *d_result4,
                               void *stream) {
  if (!d_result4 || (!d_in && length)) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  return SC_LAUNCH(sc::launch_reduce_array(d_in, length, initvalue, d_result4, as_stream(stream)), "reduce launch");
}

size_t simdcomp_hip_payload_offsets_workspace(size_t nblocks) { return sc::scan_workspace_bytes(nblocks); }

int simdcomp_hip_payload_offsets(const uint8_t *d_bits, size_t nblocks, uint64_t *d_payload_offsets,
                                 uint64_t *d_total_bytes, void *d_workspace, void *stream) {
  if (nblocks && (!d_bits || !d_payload_offsets || !d_workspace)) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  return SC_LAUNCH(sc::launch_payload_offsets(d_bits, nblocks, d_payload_offsets, d_total_bytes, d_workspace,
                                               as_stream(stream)),
                    "payload offsets launch");
}

// ---- the offsets scan for callers that cannot allocate ------------------------------------------------------------
// Same kernels as simdcomp_hip_payload_offsets; their tile sums (8 bytes per 4096 blocks) live in a device array of
// this library instead of a caller's workspace.  The array's name is only a host-side handle: its device address is
// looked up once per device (hipGetSymbolAddress, with that device current) and that address is what the kernels get.
namespace {
constexpr size_t SCRATCH_BYTES = (size_t)1 << 20;
constexpr int SCRATCH_MAX_DEVICES = 64;
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];
std::atomic<void *> g_scratch_ptr[SCRATCH_MAX_DEVICES];

// Device address of `scratch` on the current device (every GPU holds its own copy at its own address).
int scratch_on_current_device(void **out) {
  int dev = -1;
  if (hipError_t e = hipGetDevice(&dev)) return hip_result(e, "hipGetDevice");
  if (dev < 0 || dev >= SCRATCH_MAX_DEVICES) return fail(SIMDCOMP_HIP_ESIZE, "device ordinal beyond the scratch table");
  void *p = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (!p) {
    if (hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch))) return hip_result(e, "hipGetSymbolAddress(scratch)");
    if (!p) return fail(SIMDCOMP_HIP_ENULL, "hipGetSymbolAddress returned no address for the scratch array");
    g_scratch_ptr[dev].store(p, std::memory_order_release);  // threads racing here store the same address
  }
  *out = p;
  return SIMDCOMP_HIP_OK;
}
}  // namespace

size_t simdcomp_hip_payload_offsets_noalloc_max_blocks(void) {
  // scan_workspace_bytes(n) = 8 bytes per 4096 blocks
  size_t lo = 0, hi = (size_t)1 << 40;
  while (lo < hi) {
    const size_t mid = lo + (hi - lo + 1) / 2;
    if (sc::scan_workspace_bytes(mid) <= SCRATCH_BYTES) lo = mid; else hi = mid - 1;
  }
  return lo;
}

int simdcomp_hip_payload_offsets_noalloc(const uint8_t *d_bits, size_t nblocks, uint64_t *d_payload_offsets,
                                         uint64_t *d_total_bytes, void *stream) {
  if (nblocks && (!d_bits || !d_payload_offsets)) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (sc::scan_workspace_bytes(nblocks) > SCRATCH_BYTES)
    return fail(SIMDCOMP_HIP_ESIZE, "nblocks beyond simdcomp_hip_payload_offsets_noalloc_max_blocks(): use "
                                    "simdcomp_hip_payload_offsets with a workspace");
  void *ws = nullptr;
  if (int rc = scratch_on_current_device(&ws)) return rc;
  // nothing in the scratch has to start at zero: the first kernel writes every tile sum the later ones read
  return SC_LAUNCH(sc::launch_payload_offsets(d_bits, nblocks, d_payload_offsets, d_total_bytes, ws, as_stream(stream)),
                    "payload offsets launch");
}

int simdcomp_hip_pack_length(const uint32_t *d_in, size_t length, void *d_out, uint32_t bit, void *stream) {
  if (bit > 32) return fail(SIMDCOMP_HIP_EBIT, "bit width > 32");
  if (length == 0 || bit == 0) return SIMDCOMP_HIP_OK;
  if (!d_in || !d_out) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (bit == 32)
    return hip_result(hipMemcpyAsync(d_out, d_in, length * 4, hipMemcpyDeviceToDevice, as_stream(stream)), "copy");
  const size_t nfull = length / 128, rest = length % 128;
  if (int rc = simdcomp_hip_pack(d_in, d_out, nfull, bit, 1, stream)) return rc;
  if (rest) {
    if (!aligned16(d_in) || !aligned16(d_out)) return fail(SIMDCOMP_HIP_EALIGN, "pointer not 16-byte aligned");
    sc::PackArgs a{};
    a.in = d_in + nfull * 128;
    a.payload = static_cast<uint8_t *>(d_out) + nfull * 16 * bit;
    a.nblocks = 1;
    a.length = rest;
    a.bit = bit;
    return SC_LAUNCH(sc::launch_pack(sc::MODE_PLAIN, false, true, a, as_stream(stream)), "pack tail launch");
  }
  return SIMDCOMP_HIP_OK;
}

static int unpack_length_common(int mode, uint32_t initvalue, const void *d_in, size_t length, uint32_t *d_out,
                                uint32_t bit, void *stream) {
  if (bit > 32) return fail(SIMDCOMP_HIP_EBIT, "bit width > 32");
  if (length == 0) return SIMDCOMP_HIP_OK;
  if (!d_out || (!d_in && bit)) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (bit == 32)
    return hip_result(hipMemcpyAsync(d_out, d_in, length * 4, hipMemcpyDeviceToDevice, as_stream(stream)), "copy");
  if (!aligned16(d_in) || !aligned16(d_out)) return fail(SIMDCOMP_HIP_EALIGN, "pointer not 16-byte aligned");
  const size_t nfull = length / 128, rest = length % 128;
  if (int rc = do_unpack(mode, initvalue, nullptr, d_in, nullptr, nullptr, bit, d_out, nfull, stream)) return rc;
  if (rest) {
    sc::UnpackArgs a{};
    a.payload = static_cast<const uint8_t *>(d_in) + nfull * 16 * bit;
    a.out = d_out + nfull * 128;
    a.nblocks = 1;
    a.length = rest;
    a.bit = bit;
    a.seed0 = initvalue;
    return SC_LAUNCH(sc::launch_unpack(mode, true, a, as_stream(stream)), "unpack tail launch");
  }
  return SIMDCOMP_HIP_OK;
}

int simdcomp_hip_unpack_length(const void *d_in, size_t length, uint32_t *d_out, uint32_t bit, void *stream) {
  return unpack_length_common(sc::MODE_PLAIN, 0, d_in, length, d_out, bit, stream);
}

int simdcomp_hip_packFOR_length(uint32_t initvalue, const uint32_t *d_in, size_t length, void *d_out, uint32_t bit,
                                int carry_in, void *stream) {
  if (bit > 32) return fail(SIMDCOMP_HIP_EBIT, "bit width > 32");
  if (length == 0 || bit == 0) return SIMDCOMP_HIP_OK;
  if (!d_in || !d_out) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (bit == 32)
    return hip_result(hipMemcpyAsync(d_out, d_in, length * 4, hipMemcpyDeviceToDevice, as_stream(stream)), "copy");
  if (!aligned16(d_in) || !aligned16(d_out)) return fail(SIMDCOMP_HIP_EALIGN, "pointer not 16-byte aligned");
  sc::PackArgs a{};
  a.in = d_in;
  a.payload = d_out;
  a.nblocks = (length + 127) / 128;
  a.length = length;
  a.bit = bit;
  a.seed0 = initvalue;
  a.carry_blocks = 1;
  a.carry_first = carry_in != 0;
  return SC_LAUNCH(sc::launch_pack(sc::MODE_FOR, false, true, a, as_stream(stream)), "packFOR_length launch");
}

int simdcomp_hip_unpackFOR_length(uint32_t initvalue, const void *d_in, size_t length, uint32_t *d_out,
                                  uint32_t bit, void *stream) {
  return unpack_length_common(sc::MODE_FOR, initvalue, d_in, length, d_out, bit, stream);
}

size_t simdcomp_hip_chain_workspace(size_t nblocks) { return sc::chain_workspace_bytes(nblocks); }

int simdcomp_hip_chain_offsets(uint32_t initvalue, const void *d_payload, const uint64_t *d_payload_offsets,
                               const uint8_t *d_bits, size_t nblocks, int stream_layout, uint32_t *d_initvalues_out,
                               void *d_workspace, void *stream) {
  if (nblocks == 0) return SIMDCOMP_HIP_OK;
  if (!d_payload || !d_payload_offsets || !d_bits || !d_initvalues_out || !d_workspace)
    return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if ((!stream_layout && !aligned16(d_payload)) || !aligned16(d_workspace))
    return fail(SIMDCOMP_HIP_EALIGN, "pointer not 16-byte aligned");
  return SC_LAUNCH(sc::launch_chain_offsets(initvalue, d_payload, d_payload_offsets, d_bits, nblocks, stream_layout != 0,
                                             d_initvalues_out, d_workspace, as_stream(stream)),
                    "chain offsets launch");
}

int simdcomp_hip_stream_build(const void *d_payload, const uint64_t *d_payload_offsets, const uint8_t *d_bits,
                              size_t nblocks, void *d_stream, void *stream) {
  if (nblocks == 0) return SIMDCOMP_HIP_OK;
  if (!d_payload || !d_payload_offsets || !d_bits || !d_stream) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (!aligned16(d_payload)) return fail(SIMDCOMP_HIP_EALIGN, "payload arena not 16-byte aligned");
  return SC_LAUNCH(sc::launch_stream_build(d_payload, d_payload_offsets, d_bits, nblocks,
                                            static_cast<uint8_t *>(d_stream), as_stream(stream)),
                    "stream build launch");
}

int simdcomp_hip_stream_unbuild(const void *d_stream, const uint64_t *d_payload_offsets, const uint8_t *d_bits,
                                size_t nblocks, void *d_payload, void *stream) {
  if (nblocks == 0) return SIMDCOMP_HIP_OK;
  if (!d_payload || !d_payload_offsets || !d_bits || !d_stream) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (!aligned16(d_payload)) return fail(SIMDCOMP_HIP_EALIGN, "payload arena not 16-byte aligned");
  return SC_LAUNCH(sc::launch_stream_unbuild(static_cast<const uint8_t *>(d_stream), d_payload_offsets, d_bits, nblocks,
                                              d_payload, as_stream(stream)),
                    "stream unbuild launch");
}

size_t simdcomp_hip_stream_index_workspace(size_t stream_bytes) { return sc::stream_index_workspace_bytes(stream_bytes); }

int simdcomp_hip_stream_index(const void *d_stream, size_t stream_bytes, size_t max_blocks, uint8_t *d_bits,
                              uint64_t *d_payload_offsets, uint64_t *d_result, void *d_workspace, void *stream) {
  if (!d_result) return fail(SIMDCOMP_HIP_ENULL, "result pointer is NULL");
  if (stream_bytes && (!d_stream || !d_workspace)) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (stream_bytes && max_blocks && (!d_bits || !d_payload_offsets)) return fail(SIMDCOMP_HIP_ENULL, "NULL output array");
  if (!aligned16(d_workspace)) return fail(SIMDCOMP_HIP_EALIGN, "workspace not 16-byte aligned");
  return SC_LAUNCH(sc::launch_stream_index(static_cast<const uint8_t *>(d_stream), stream_bytes, max_blocks, d_bits,
                                            d_payload_offsets, d_result, d_workspace, /*clear_tail=*/false, as_stream(stream)),
                    "stream index launch");
}

// ---- example.c's compress() and decode loops as one call each ----------------------------------------------------
namespace {
size_t up256(size_t x) { return (x + 255) / 256 * 256; }
}  // namespace

size_t simdcomp_hip_stream_compress_workspace(size_t nblocks) {
  return up256(nblocks) + up256(8 * nblocks) + 256 + up256(sc::scan_workspace_bytes(nblocks)) + 256;
}

int simdcomp_hip_stream_compress(int mode, uint32_t initvalue, const uint32_t *d_in, size_t nblocks, void *d_stream,
                                 uint64_t *d_stream_bytes, void *d_workspace, void *stream) {
  if (mode != SIMDCOMP_HIP_D1 && mode != SIMDCOMP_HIP_PLAIN) return fail(SIMDCOMP_HIP_EMODE, "mode must be D1 or PLAIN");
  if (!d_stream_bytes) return fail(SIMDCOMP_HIP_ENULL, "d_stream_bytes is NULL");
  if (nblocks == 0) return hip_result(hipMemsetAsync(d_stream_bytes, 0, sizeof(uint64_t), as_stream(stream)), "memset");
  if (!d_in || !d_stream || !d_workspace) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (!aligned16(d_in) || !aligned16(d_workspace)) return fail(SIMDCOMP_HIP_EALIGN, "pointer not 16-byte aligned");
  uint8_t *ws = static_cast<uint8_t *>(d_workspace);
  uint8_t *bits = ws;
  uint64_t *off = reinterpret_cast<uint64_t *>(ws + up256(nblocks));
  uint64_t *total = reinterpret_cast<uint64_t *>(ws + up256(nblocks) + up256(8 * nblocks));
  void *scan_ws = ws + up256(nblocks) + up256(8 * nblocks) + 256;
  int rc = mode == SIMDCOMP_HIP_D1 ? simdcomp_hip_maxbitsd1(initvalue, nullptr, d_in, nblocks, bits, nullptr, stream)
                                   : simdcomp_hip_maxbits(d_in, nblocks, bits, stream);
  if (rc) return rc;
  if ((rc = simdcomp_hip_payload_offsets(bits, nblocks, off, total, scan_ws, stream))) return rc;
  if ((rc = simdcomp_hip_pack_stream(mode, 0, initvalue, nullptr, d_in, d_stream, off, bits, nblocks, nullptr, stream))) return rc;
  return SC_LAUNCH(sc::launch_stream_bytes(total, nblocks, d_stream_bytes, as_stream(stream)), "stream bytes launch");
}

size_t simdcomp_hip_stream_decode_workspace(size_t stream_bytes, size_t nblocks) {
  return up256(nblocks) + up256(8 * nblocks) + up256(4 * nblocks) + up256(sc::stream_index_workspace_bytes(stream_bytes)) +
         up256(sc::chain_workspace_bytes(nblocks)) + 256;
}

int simdcomp_hip_stream_decode(int mode, uint32_t initvalue, const void *d_stream, size_t stream_bytes, size_t nblocks,
                               uint32_t *d_out, uint64_t *d_result, void *d_workspace, void *stream) {
  if (mode != SIMDCOMP_HIP_D1 && mode != SIMDCOMP_HIP_PLAIN) return fail(SIMDCOMP_HIP_EMODE, "mode must be D1 or PLAIN");
  if (!d_result) return fail(SIMDCOMP_HIP_ENULL, "d_result is NULL");
  if (nblocks == 0) return hip_result(hipMemsetAsync(d_result, 0, 3 * sizeof(uint64_t), as_stream(stream)), "memset");
  if (!d_stream || !d_out || !d_workspace) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (!aligned16(d_out) || !aligned16(d_workspace)) return fail(SIMDCOMP_HIP_EALIGN, "pointer not 16-byte aligned");
  uint8_t *ws = static_cast<uint8_t *>(d_workspace);
  uint8_t *bits = ws;
  uint64_t *off = reinterpret_cast<uint64_t *>(ws + up256(nblocks));
  uint32_t *seeds = reinterpret_cast<uint32_t *>(ws + up256(nblocks) + up256(8 * nblocks));
  uint8_t *index_ws = ws + up256(nblocks) + up256(8 * nblocks) + up256(4 * nblocks);
  uint8_t *chain_ws = index_ws + up256(sc::stream_index_workspace_bytes(stream_bytes));
  // a stream with fewer records than nblocks leaves the rest as zero-width blocks at offset 0 (d_result[0] tells): the
  // index clears exactly those entries (all of them before it writes what it finds, when the stream is bad)
  int rc = SC_LAUNCH(sc::launch_stream_index(static_cast<const uint8_t *>(d_stream), stream_bytes, nblocks, bits, off, d_result,
                                              index_ws, /*clear_tail=*/true, as_stream(stream)),
                      "stream index launch");
  if (rc) return rc;
  if (mode == SIMDCOMP_HIP_D1) {
    if ((rc = simdcomp_hip_chain_offsets(initvalue, d_stream, off, bits, nblocks, 1, seeds, chain_ws, stream))) return rc;
    return simdcomp_hip_unpack_stream(SIMDCOMP_HIP_D1, seeds, d_stream, off, bits, d_out, nblocks, stream);
  }
  return simdcomp_hip_unpack_stream(SIMDCOMP_HIP_PLAIN, nullptr, d_stream, off, bits, d_out, nblocks, stream);
}

namespace {
int do_query(int mode, int what, uint32_t initvalue, const uint32_t *d_initvalues, const void *d_payload,
             const uint64_t *d_payload_offsets, const uint8_t *d_bits, uint32_t bit, int stream_layout, sc::QueryArgs a,
             void *stream) {
  if (mode < 0 || mode > 2) return fail(SIMDCOMP_HIP_EMODE, "unknown mode");
  if (a.nqueries == 0) return SIMDCOMP_HIP_OK;
  if (!d_bits && bit > 32) return fail(SIMDCOMP_HIP_EBIT, "bit width > 32");
  if (!d_payload && (d_bits || bit)) return fail(SIMDCOMP_HIP_ENULL, "payload pointer is NULL");
  if (d_bits && !d_payload_offsets) return fail(SIMDCOMP_HIP_ENULL, "per-block widths need per-block payload offsets");
  if (stream_layout && !d_bits) return fail(SIMDCOMP_HIP_ENULL, "stream layout needs per-block widths");
  if (!stream_layout && !aligned16(d_payload)) return fail(SIMDCOMP_HIP_EALIGN, "payload not 16-byte aligned");
  a.payload = d_payload;
  a.off = d_bits ? d_payload_offsets : nullptr;
  a.bits = d_bits;
  a.seeds = d_initvalues;
  a.bit = bit;
  a.seed0 = initvalue;
  a.stream_layout = stream_layout ? 1u : 0u;
  return SC_LAUNCH(sc::launch_query(mode, what, a, as_stream(stream)), "query launch");
}
}  // namespace

int simdcomp_hip_select(int mode, uint32_t initvalue, const uint32_t *d_initvalues, const void *d_payload,
                        const uint64_t *d_payload_offsets, const uint8_t *d_bits, uint32_t bit, int stream_layout,
                        const uint64_t *d_query_blocks, const int32_t *d_slots, size_t nqueries,
                        uint32_t *d_values_out, void *stream) {
  if (nqueries && (!d_slots || !d_values_out)) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  sc::QueryArgs a{};
  a.query_blocks = d_query_blocks;
  a.slots = d_slots;
  a.nqueries = nqueries;
  a.values_out = d_values_out;
  return do_query(mode, sc::Q_SELECT, initvalue, d_initvalues, d_payload, d_payload_offsets, d_bits, bit, stream_layout, a,
                  stream);
}

int simdcomp_hip_search(int mode, uint32_t initvalue, const uint32_t *d_initvalues, const void *d_payload,
                        const uint64_t *d_payload_offsets, const uint8_t *d_bits, uint32_t bit, int stream_layout,
                        const uint64_t *d_query_blocks, const uint32_t *d_keys, const int32_t *d_lengths,
                        size_t nqueries, int32_t *d_positions_out, uint32_t *d_values_out, uint32_t *d_last4_out,
                        void *stream) {
  if (nqueries && (!d_keys || !d_values_out || !d_positions_out)) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (d_last4_out && mode != SIMDCOMP_HIP_D1) return fail(SIMDCOMP_HIP_EMODE, "the offset vector exists for d1 only");
  sc::QueryArgs a{};
  a.query_blocks = d_query_blocks;
  a.keys = d_keys;
  a.lengths = d_lengths;
  a.nqueries = nqueries;
  a.values_out = d_values_out;
  a.positions_out = d_positions_out;
  a.last4_out = d_last4_out;
  return do_query(mode, sc::Q_SEARCH, initvalue, d_initvalues, d_payload, d_payload_offsets, d_bits, bit, stream_layout, a,
                  stream);
}

size_t simdcomp_hip_block_directory_bytes(size_t nblocks) { return nblocks * sc::DIRECTORY_ENTRY_BYTES; }

int simdcomp_hip_block_directory(uint32_t initvalue, const uint32_t *d_initvalues, const uint64_t *d_payload_offsets,
                                 const uint8_t *d_bits, size_t nblocks, int stream_layout, void *d_directory, void *stream) {
  if (nblocks == 0) return SIMDCOMP_HIP_OK;
  if (!d_payload_offsets || !d_bits || !d_directory) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (!aligned16(d_directory)) return fail(SIMDCOMP_HIP_EALIGN, "directory not 16-byte aligned");
  return SC_LAUNCH(sc::launch_block_directory(initvalue, d_initvalues, d_payload_offsets, d_bits, nblocks, stream_layout ? 1u : 0u,
                                              d_directory, as_stream(stream)),
                   "block directory launch");
}

namespace {
int do_directory_query(int mode, int what, const void *d_payload, const void *d_directory, sc::QueryArgs a, void *stream) {
  if (mode < 0 || mode > 2) return fail(SIMDCOMP_HIP_EMODE, "unknown mode");
  if (a.nqueries == 0) return SIMDCOMP_HIP_OK;
  if (!d_payload || !d_directory) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (!aligned16(d_directory)) return fail(SIMDCOMP_HIP_EALIGN, "directory not 16-byte aligned");
  a.payload = d_payload;
  a.directory = d_directory;
  return SC_LAUNCH(sc::launch_query(mode, what, a, as_stream(stream)), "query launch");
}
}  // namespace

int simdcomp_hip_select_directory(int mode, const void *d_payload, const void *d_directory, const uint64_t *d_query_blocks,
                                  const int32_t *d_slots, size_t nqueries, uint32_t *d_values_out, void *stream) {
  if (nqueries && (!d_slots || !d_values_out)) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  sc::QueryArgs a{};
  a.query_blocks = d_query_blocks;
  a.slots = d_slots;
  a.nqueries = nqueries;
  a.values_out = d_values_out;
  return do_directory_query(mode, sc::Q_SELECT, d_payload, d_directory, a, stream);
}

int simdcomp_hip_search_directory(int mode, const void *d_payload, const void *d_directory, const uint64_t *d_query_blocks,
                                  const uint32_t *d_keys, const int32_t *d_lengths, size_t nqueries, int32_t *d_positions_out,
                                  uint32_t *d_values_out, uint32_t *d_last4_out, void *stream) {
  if (nqueries && (!d_keys || !d_values_out || !d_positions_out)) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (d_last4_out && mode != SIMDCOMP_HIP_D1) return fail(SIMDCOMP_HIP_EMODE, "the offset vector exists for d1 only");
  sc::QueryArgs a{};
  a.query_blocks = d_query_blocks;
  a.keys = d_keys;
  a.lengths = d_lengths;
  a.nqueries = nqueries;
  a.values_out = d_values_out;
  a.positions_out = d_positions_out;
  a.last4_out = d_last4_out;
  return do_directory_query(mode, sc::Q_SEARCH, d_payload, d_directory, a, stream);
}

int simdcomp_hip_scand1(uint32_t initvalue, const uint32_t *d_initvalues, const void *d_payload,
                        const uint64_t *d_payload_offsets, const uint8_t *d_bits, uint32_t bit, int stream_layout,
                        size_t nblocks, uint32_t *d_last4_out, void *stream) {
  if (nblocks && !d_last4_out) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  sc::QueryArgs a{};
  a.nqueries = nblocks;
  a.last4_out = d_last4_out;
  return do_query(SIMDCOMP_HIP_D1, sc::Q_SCAN, initvalue, d_initvalues, d_payload, d_payload_offsets, d_bits, bit,
                  stream_layout, a, stream);
}

namespace {
int lanes_ok(int lanes) { return lanes == 4 || lanes == 8 || lanes == 16; }
}  // namespace

int simdcomp_hip_pack_lanes(int lanes, const uint32_t *d_in, void *d_out, size_t nblocks, uint32_t bit, int masked,
                            void *stream) {
  if (!lanes_ok(lanes)) return fail(SIMDCOMP_HIP_EMODE, "lanes must be 4, 8 or 16");
  if (int rc = check_common(d_in, d_out, nblocks, bit, bit != 0)) return rc;
  sc::PackArgs a{};
  a.in = d_in;
  a.payload = d_out;
  a.lane_groups = (uint32_t)lanes / 4;
  a.nblocks = nblocks * a.lane_groups;
  a.length = a.nblocks * 128;
  a.bit = bit;
  return SC_LAUNCH(sc::launch_pack(sc::MODE_PLAIN, masked != 0, false, a, as_stream(stream)), "pack launch");
}

int simdcomp_hip_unpack_lanes(int lanes, const void *d_in, uint32_t *d_out, size_t nblocks, uint32_t bit, void *stream) {
  if (!lanes_ok(lanes)) return fail(SIMDCOMP_HIP_EMODE, "lanes must be 4, 8 or 16");
  if (int rc = check_common(d_out, d_in, nblocks, bit, bit != 0)) return rc;
  sc::UnpackArgs a{};
  a.payload = d_in;
  a.out = d_out;
  a.lane_groups = (uint32_t)lanes / 4;
  a.nblocks = nblocks * a.lane_groups;
  a.length = a.nblocks * 128;
  a.bit = bit;
  return SC_LAUNCH(sc::launch_unpack(sc::MODE_PLAIN, false, a, as_stream(stream)), "unpack launch");
}

int simdcomp_hip_maxbits_lanes(int lanes, const uint32_t *d_in, size_t nblocks, uint8_t *d_bits, void *stream) {
  if (!lanes_ok(lanes)) return fail(SIMDCOMP_HIP_EMODE, "lanes must be 4, 8 or 16");
  if (nblocks == 0) return SIMDCOMP_HIP_OK;
  if (!d_in || !d_bits) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  if (!aligned16(d_in)) return fail(SIMDCOMP_HIP_EALIGN, "pointer not 16-byte aligned");
  sc::ReduceArgs a{};
  a.group = (uint32_t)lanes / 4;
  a.in = d_in;
  a.bits_out = d_bits;
  a.nblocks = nblocks;
  return SC_LAUNCH(sc::launch_reduce(sc::RED_MAXBITS, a, as_stream(stream)), "maxbits launch");
}

int simdcomp_hip_checksum(const uint32_t *d_data, const uint32_t *d_other, size_t nwords, uint64_t index0,
                          uint64_t *d_acc, void *stream) {
  if (nwords == 0) return SIMDCOMP_HIP_OK;
  if (!d_data || !d_acc) return fail(SIMDCOMP_HIP_ENULL, "NULL pointer");
  return SC_LAUNCH(sc::launch_checksum(d_data, d_other, nwords, index0, d_acc, as_stream(stream)), "checksum launch");
}

}  // extern "C"
