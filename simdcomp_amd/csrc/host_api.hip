// Reference-signature entry points (include/simdcomp.h): HOST pointers in and
// out, the work done by the HIP kernels.  Operands are staged into device
// buffers owned by this file, the same kernels the batched device entry points
// use are launched on an internal stream, and results are copied back before
// the call returns.  There is no CPU implementation behind these routines:
// without a working GPU they print the HIP error and abort.
//
// Long arrays (*_length): when both caller buffers are pinned host memory that
// the GPU can address (hipHostMalloc / hipHostRegister) and 16-byte aligned, the
// kernels run on them in place -- one launch reads the input and writes the
// result over the host link, both directions at once, no staging.  Otherwise the
// array moves in pieces of PIECE integers through three sets of staging buffers
// on three streams -- uploads, kernels, downloads -- chained by events, so
// that the link carries both directions all the time (one stream per piece
// with upload, kernel and download in a row leaves each direction idle half
// of the time: 7.0 against 10.9 G int32/s at 32 bits).  Pageable caller memory
// works too (the HIP runtime stages it).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <string.h>
#include <vector>

#include "../../include/simdcomp.h"
#include "../../include/simdcomp_hip.h"
#include "codec_kernels.h"

namespace {

constexpr size_t PIECE = (size_t)1 << 24;  // integers per pipeline piece (64 MiB), a multiple of 128

std::atomic<simdcomp_hip_error_handler> g_handler{nullptr};
std::atomic<void *> g_handler_arg{nullptr};

// The reference's signatures have no error channel.  A failure of the GPU path goes to the handler the program
// installed (which must not return: longjmp, exit, or a C++ throw through a C++ caller), else to stderr + abort.
[[noreturn]] void die(const char *where, const char *what) {
  char msg[320];
  snprintf(msg, sizeof msg, "simdcomp_hip: %s failed: %s", where, what);
  if (simdcomp_hip_error_handler h = g_handler.load()) {
    // the handler may leave through longjmp: nothing of this call may still be writing into the caller's buffers
    // then (copies of the piece pipeline that are in flight), so the device is drained first; errors are ignored
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    h(msg, g_handler_arg.load());
  }
  fprintf(stderr, "%s\n", msg);
  abort();
}

void ck(hipError_t e, const char *where) {
  if (e != hipSuccess) die(where, hipGetErrorString(e));
}

void ckapi(int rc, const char *where) {
  if (rc != 0) die(where, simdcomp_hip_last_error());
}

constexpr size_t ZC_BYTES = 64 << 10;  // operands up to this size take the mapped-host-buffer path
constexpr int MAX_DEVICES = 64;

// Everything one host-API call needs on one device: two streams, two sets of staging buffers, the mapped host
// buffers of the small-operand path.  A context is used by one thread at a time and never shared, so calls from
// different threads run side by side like the reference's (which is stateless, src/simdbitpacking.c:14379-14399);
// when a thread ends its contexts go back to a pool and serve the next new thread.  Contexts are never destroyed:
// no HIP call is made from a destructor that may run while the process is exiting.
constexpr int SLOTS = 3;
enum { S_UP = 0, S_KERNEL = 1, S_DOWN = 2 };  // stream roles of the pipeline; small calls use stream 0 alone

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  __asm__ __volatile__("yield");
#endif
}

struct Staging {
  int device = 0;
  hipStream_t stream[3] = {nullptr, nullptr, nullptr};
  void *in[SLOTS] = {nullptr, nullptr, nullptr};
  void *out[SLOTS] = {nullptr, nullptr, nullptr};
  size_t in_cap[SLOTS] = {0, 0, 0}, out_cap[SLOTS] = {0, 0, 0};
  hipEvent_t up_done[SLOTS], kernel_done[SLOTS], down_done[SLOTS];
  bool can_poll = false;  // the device can write a word of host memory from a stream
  uint32_t *result4 = nullptr;  // device scratch for whole-array reductions
  // Small operands skip the copies: two host buffers that the GPU maps; the kernels read and write them in place
  // over PCIe and the call costs one launch and one synchronisation instead of two copies on top of those.
  uint8_t *zin = nullptr, *zout = nullptr;
  // Completion of a small call: the stream writes a sequence number into a word of mapped host memory behind the
  // kernel, and the calling thread polls that word instead of going through hipStreamSynchronize.
  uint32_t *done_word = nullptr;
  uint32_t done_seq = 0;

  // with `device` current
  void create(int dev) {
    device = dev;
    for (int i = 0; i < 3; ++i) ck(hipStreamCreateWithFlags(&stream[i], hipStreamNonBlocking), "hipStreamCreate");
    for (int i = 0; i < SLOTS; ++i) {
      ck(hipEventCreateWithFlags(&up_done[i], hipEventDisableTiming), "hipEventCreate");
      ck(hipEventCreateWithFlags(&kernel_done[i], hipEventDisableTiming), "hipEventCreate");
      ck(hipEventCreateWithFlags(&down_done[i], hipEventDisableTiming), "hipEventCreate");
    }
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeCanUseStreamWaitValue, dev) == hipSuccess) can_poll = v != 0;
    else (void)hipGetLastError();
    ck(hipMalloc(reinterpret_cast<void **>(&result4), 64), "hipMalloc");
    ck(hipHostMalloc(reinterpret_cast<void **>(&zin), ZC_BYTES + 128, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc");
    ck(hipHostMalloc(reinterpret_cast<void **>(&zout), ZC_BYTES + 128, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc");
    ck(hipHostMalloc(reinterpret_cast<void **>(&done_word), 64, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc");
    *done_word = 0;
  }

  // Wait until everything enqueued on `st` so far has finished and its writes to the mapped buffers are visible.
  void finish(hipStream_t st) {
    static const bool poll = [] {
      const char *e = getenv("SIMDCOMP_HIP_POLL");
      return !(e && e[0] == '0');
    }();
    if (poll && can_poll) {
      const uint32_t want = ++done_seq;
      if (hipStreamWriteValue32(st, done_word, want, 0) == hipSuccess) {
        // A small call on an idle GPU completes in about ten microseconds: poll for a hundred, then leave the
        // core to others and wait in the runtime (a busy or shared GPU, a fault on the device).
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0;; ++spins) {
          if (__atomic_load_n(done_word, __ATOMIC_ACQUIRE) == want) {
            // every so often through the runtime as well, so that it retires the commands it keeps track of
            if ((want & 1023u) == 0) ck(hipStreamSynchronize(st), "hipStreamSynchronize");
            return;
          }
          cpu_relax();
          if ((spins & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(100)) break;
        }
      } else {
        (void)hipGetLastError();
        can_poll = false;
      }
    }
    ck(hipStreamSynchronize(st), "hipStreamSynchronize");
  }

  // Staging buffers of one slot.  Called with the context's streams idle (every host-API call drains them
  // before it returns).
  void reserve(int slot, size_t in_bytes, size_t out_bytes) {
    in_bytes += 64;  // 16 bytes in front of din() hold the previous row of a piecewise FOR pack
    out_bytes += 64;
    if (in_bytes > in_cap[slot]) {
      if (in[slot]) ck(hipFree(in[slot]), "hipFree");
      in[slot] = nullptr;
      in_cap[slot] = 0;
      ck(hipMalloc(&in[slot], in_bytes), "hipMalloc");
      in_cap[slot] = in_bytes;
    }
    if (out_bytes > out_cap[slot]) {
      if (out[slot]) ck(hipFree(out[slot]), "hipFree");
      out[slot] = nullptr;
      out_cap[slot] = 0;
      ck(hipMalloc(&out[slot], out_bytes), "hipMalloc");
      out_cap[slot] = out_bytes;
    }
  }
  void drain() {
    for (int i = 0; i < 3; ++i) ck(hipStreamSynchronize(stream[i]), "hipStreamSynchronize");
  }
  uint8_t *zi() { return zin + 16; }  // 16 bytes in front, as for din()
  uint8_t *din(int slot) { return static_cast<uint8_t *>(in[slot]) + 16; }
  uint8_t *dout(int slot) { return static_cast<uint8_t *>(out[slot]); }
};

struct Pool {
  std::mutex mu;
  std::vector<Staging *> idle[MAX_DEVICES];
  std::vector<int> devices;  // SIMDCOMP_HIP_DEVICES, empty: the calling thread's current device
  std::atomic<unsigned> next_thread{0};

  // What went wrong while the pool was set up, reported by the first call that needs it: the constructor runs
  // inside a function-local static's initialisation, and a handler that leaves through longjmp from there would
  // leave that initialisation locked for ever.
  const char *bad_where = nullptr, *bad_what = nullptr;

  Pool() { discover(); }
  void discover() {
    int n = 0;
    const hipError_t e0 = hipGetDeviceCount(&n);
    if (e0 != hipSuccess) return fail("hipGetDeviceCount", hipGetErrorString(e0));
    if (n <= 0) return fail("device discovery", "no HIP device");
    // SIMDCOMP_HIP_DEVICES=0,2,5: host-API calls are spread over these GPUs, one per calling thread (threads take
    // them in turn, in the order of their first call); unset: every call runs on the caller's current HIP device.
    if (const char *e = getenv("SIMDCOMP_HIP_DEVICES")) {
      const char *p = e;
      while (*p) {
        char *q = nullptr;
        const long v = strtol(p, &q, 10);
        if (q == p) return fail("SIMDCOMP_HIP_DEVICES", "expected a comma-separated list of device ordinals");
        if (v < 0 || v >= n || v >= MAX_DEVICES) return fail("SIMDCOMP_HIP_DEVICES", "device ordinal out of range");
        devices.push_back((int)v);
        p = q;
        while (*p == ',' || *p == ' ') ++p;
      }
    }
  }
  void fail(const char *where, const char *what) {
    bad_where = where;
    bad_what = what;
  }
};

Pool &pool() {
  static Pool *p = new Pool;  // never destroyed (see Staging); its constructor cannot fail
  return *p;
}

// The contexts the current thread holds, one per device it has used; handed back when the thread ends.
struct Lease {
  Staging *held[MAX_DEVICES] = {};
  int assigned = -1;  // device taken from SIMDCOMP_HIP_DEVICES
  ~Lease() {
    bool any = false;
    for (int d = 0; d < MAX_DEVICES; ++d) any = any || held[d];
    if (!any) return;  // (a thread that never called the host API must not create the pool on its way out)
    Pool &p = pool();
    std::lock_guard<std::mutex> lock(p.mu);
    for (int d = 0; d < MAX_DEVICES; ++d)
      if (held[d]) p.idle[d].push_back(held[d]);
  }
};
thread_local Lease t_lease;

// Scope of one host-API call: picks the device, makes it current for the calling thread (and puts the caller's
// device back afterwards), and provides that thread's context on it.
struct Call {
  int prev = -1, dev = -1;
  Staging *ctx = nullptr;
  Call() {
    Pool &p = pool();
    if (p.bad_where) die(p.bad_where, p.bad_what);
    // HIP keeps a thread's last error until it is read: a failure of the application's own before this call must
    // not surface as a failed launch of ours
    (void)hipGetLastError();
    ck(hipGetDevice(&prev), "hipGetDevice");
    if (p.devices.empty()) {
      dev = prev;
    } else {
      if (t_lease.assigned < 0) t_lease.assigned = p.devices[p.next_thread.fetch_add(1) % p.devices.size()];
      dev = t_lease.assigned;
    }
    if (dev < 0 || dev >= MAX_DEVICES) die("device selection", "device ordinal beyond the context table");
    if (dev != prev) ck(hipSetDevice(dev), "hipSetDevice");
    ctx = t_lease.held[dev];
    if (!ctx) {
      {
        std::lock_guard<std::mutex> lock(p.mu);
        if (!p.idle[dev].empty()) {
          ctx = p.idle[dev].back();
          p.idle[dev].pop_back();
        }
      }
      if (!ctx) {
        ctx = new Staging;
        ctx->create(dev);
      }
      t_lease.held[dev] = ctx;
    }
  }
  ~Call() {
    if (dev != prev) (void)hipSetDevice(prev);
  }
  Call(const Call &) = delete;
  Call &operator=(const Call &) = delete;
};

void h2d(void *dst, const void *src, size_t n, hipStream_t st) {
  if (n) ck(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, st), "hipMemcpyAsync H2D");
}
void d2h(void *dst, const void *src, size_t n, hipStream_t st) {
  if (n) ck(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, st), "hipMemcpyAsync D2H");
}

// ---- one 128-integer block ---------------------------------------------------

// groups = 1: the 4-lane block of 128 integers; 2 / 4: one 8- / 16-lane block of 256 / 512 integers (plain mode)
void block_pack(int mode, bool masked, uint32_t initvalue, const uint32_t *in, void *out, uint32_t bit, uint32_t groups = 1) {
  if (bit > 32) return;  // the reference's switch falls through and does nothing
  Call call;
  Staging &s = *call.ctx;
  hipStream_t st = s.stream[0];
  memcpy(s.zi(), in, 512 * (size_t)groups);
  sc::PackArgs a{};
  a.in = reinterpret_cast<const uint32_t *>(s.zi());
  a.payload = s.zout;
  a.nblocks = groups;
  a.length = 128 * (size_t)groups;
  a.bit = bit;
  a.seed0 = initvalue;  // seeds == nullptr: block 0 of a chained d1 stream / uniform frame base
  a.lane_groups = groups;
  ck(sc::launch_pack(mode, masked, false, a, st), "pack launch");
  s.finish(st);
  memcpy(out, s.zout, 16 * (size_t)bit * groups);
}

void block_unpack(int mode, uint32_t initvalue, const void *in, uint32_t *out, uint32_t bit, uint32_t groups = 1) {
  if (bit > 32) return;
  Call call;
  Staging &s = *call.ctx;
  hipStream_t st = s.stream[0];
  memcpy(s.zi(), in, 16 * (size_t)bit * groups);
  sc::UnpackArgs a{};
  a.payload = s.zi();
  a.out = reinterpret_cast<uint32_t *>(s.zout);
  a.nblocks = groups;
  a.length = 128 * (size_t)groups;
  a.bit = bit;
  a.seed0 = initvalue;
  a.lane_groups = groups;
  ck(sc::launch_unpack(mode, false, a, st), "unpack launch");
  s.finish(st);
  memcpy(out, s.zout, 512 * (size_t)groups);
}

// ---- whole-array reductions ----------------------------------------------------

void reduce_host(const uint32_t *in, size_t length, uint32_t initvalue, uint32_t r[4]) {
  uint32_t acc[4] = {0, 0, 0xFFFFFFFFu, 0};  // what an empty range yields (src/simdcomputil.c:135-136)
  uint32_t prev = initvalue;
  if (length) {
    Call call;
    Staging &s = *call.ctx;
    hipStream_t st = s.stream[0];
    if (length * 4 <= ZC_BYTES) {
      // the input is read in place; the four results are accumulated with device atomics and stay in device memory
      memcpy(s.zi(), in, length * 4);
      ckapi(simdcomp_hip_reduce_length(reinterpret_cast<const uint32_t *>(s.zi()), length, initvalue, s.result4, st), "reduce");
      d2h(r, s.result4, 16, st);
      ck(hipStreamSynchronize(st), "hipStreamSynchronize");
      return;
    }
    for (size_t done = 0; done < length; done += PIECE) {
      const size_t n = length - done < PIECE ? length - done : PIECE;
      s.reserve(0, n * 4, 0);
      h2d(s.din(0), in + done, n * 4, st);
      ckapi(simdcomp_hip_reduce_length(reinterpret_cast<const uint32_t *>(s.din(0)), n, prev, s.result4, st),
            "reduce");
      uint32_t part[4];
      d2h(part, s.result4, 16, st);
      ck(hipStreamSynchronize(st), "hipStreamSynchronize");
      acc[0] |= part[0];
      acc[1] |= part[1];
      if (part[2] < acc[2]) acc[2] = part[2];
      if (part[3] > acc[3]) acc[3] = part[3];
      prev = in[done + n - 1];
    }
  }
  for (int i = 0; i < 4; ++i) r[i] = acc[i];
}

uint32_t width_of(uint32_t v) { return v ? 32u - (uint32_t)__builtin_clz(v) : 0u; }

// ---- arbitrary lengths ---------------------------------------------------------

// The device-side address of a caller buffer when the GPU can reach it where it lies: pinned host memory from
// hipHostMalloc or hipHostRegister (or device memory), 16-byte aligned.  nullptr for ordinary pageable memory.
void *device_view(const void *p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory is reported as an error
    return nullptr;
  }
  if (at.type != hipMemoryTypeHost && at.type != hipMemoryTypeDevice) return nullptr;
  void *d = at.devicePointer;
  if (!d || (reinterpret_cast<uintptr_t>(d) & 15u)) return nullptr;
  return d;
}

// The three-stream pipeline over the pieces of a long array.  upload(slot, first integer, count), launch(...) and
// download(...) enqueue on the stream they are given.  Slot q's input buffer is free again when the kernel that
// read it is done, its output buffer when the download that read it is done.
template <typename Up, typename Launch, typename Down>
void run_pipeline(Staging &s, size_t length, size_t in_bytes_max, size_t out_bytes_max, Up upload, Launch launch, Down download) {
  const size_t npieces = (length + PIECE - 1) / PIECE;
  const int slots = npieces < (size_t)SLOTS ? (int)npieces : SLOTS;
  for (int q = 0; q < slots; ++q) s.reserve(q, in_bytes_max, out_bytes_max);
  size_t i = 0;
  for (size_t done = 0; done < length; done += PIECE, ++i) {
    const int q = (int)(i % SLOTS);
    const size_t n = length - done < PIECE ? length - done : PIECE;
    if (i >= (size_t)SLOTS) ck(hipStreamWaitEvent(s.stream[S_UP], s.kernel_done[q], 0), "hipStreamWaitEvent");
    upload(q, done, n, s.stream[S_UP]);
    ck(hipEventRecord(s.up_done[q], s.stream[S_UP]), "hipEventRecord");
    ck(hipStreamWaitEvent(s.stream[S_KERNEL], s.up_done[q], 0), "hipStreamWaitEvent");
    if (i >= (size_t)SLOTS) ck(hipStreamWaitEvent(s.stream[S_KERNEL], s.down_done[q], 0), "hipStreamWaitEvent");
    launch(q, done, n, s.stream[S_KERNEL]);
    ck(hipEventRecord(s.kernel_done[q], s.stream[S_KERNEL]), "hipEventRecord");
    ck(hipStreamWaitEvent(s.stream[S_DOWN], s.kernel_done[q], 0), "hipStreamWaitEvent");
    download(q, done, n, s.stream[S_DOWN]);
    ck(hipEventRecord(s.down_done[q], s.stream[S_DOWN]), "hipEventRecord");
  }
  s.drain();
}

// kind: 0 simdpack_length, 1 simdpackFOR_length
void *pack_length_host(int kind, uint32_t initvalue, const uint32_t *in, size_t length, void *out, uint32_t bit) {
  if (bit > 32 || bit == 0) return out;
  uint8_t *o = static_cast<uint8_t *>(out);
  Call call;
  Staging &s = *call.ctx;
  const size_t total = simdcomp_hip_compressedbytes(length, bit);
  if (length * 4 <= ZC_BYTES) {
    hipStream_t st = s.stream[0];
    memcpy(s.zi(), in, length * 4);
    const uint32_t *z_in = reinterpret_cast<const uint32_t *>(s.zi());
    if (kind == 0) ckapi(simdcomp_hip_pack_length(z_in, length, s.zout, bit, st), "pack_length");
    else ckapi(simdcomp_hip_packFOR_length(initvalue, z_in, length, s.zout, bit, 0, st), "packFOR_length");
    s.finish(st);
    memcpy(o, s.zout, total);
    return o + total;
  }
  if (void *din = device_view(in)) {
    if (void *dout = device_view(out)) {  // in place on pinned memory: one launch, both directions of the link at once
      hipStream_t st = s.stream[0];
      const uint32_t *d_in = static_cast<const uint32_t *>(din);
      if (kind == 0) ckapi(simdcomp_hip_pack_length(d_in, length, dout, bit, st), "pack_length");
      else ckapi(simdcomp_hip_packFOR_length(initvalue, d_in, length, dout, bit, 0, st), "packFOR_length");
      ck(hipStreamSynchronize(st), "hipStreamSynchronize");
      return o + total;
    }
  }
  run_pipeline(
      s, length, PIECE * 4, simdcomp_hip_compressedbytes(PIECE, bit),
      [&](int q, size_t done, size_t n, hipStream_t st) {
        // a piece of a FOR array behind the first brings the row in front of it along (the generic loop's hand-over)
        if (kind == 1 && done > 0) h2d(s.din(q) - 16, in + done - 4, 16 + n * 4, st);
        else h2d(s.din(q), in + done, n * 4, st);
      },
      [&](int q, size_t done, size_t n, hipStream_t st) {
        const uint32_t *d_in = reinterpret_cast<const uint32_t *>(s.din(q));
        if (kind == 0) ckapi(simdcomp_hip_pack_length(d_in, n, s.dout(q), bit, st), "pack_length");
        else ckapi(simdcomp_hip_packFOR_length(initvalue, d_in, n, s.dout(q), bit, done > 0, st), "packFOR_length");
      },
      [&](int q, size_t done, size_t n, hipStream_t st) {
        d2h(o + (done / 128) * 16 * bit, s.dout(q), simdcomp_hip_compressedbytes(n, bit), st);
      });
  return o + total;
}

// kind: 0 simdunpack_length, 1 simdunpackFOR_length
const void *unpack_length_host(int kind, uint32_t initvalue, const void *in, size_t length, uint32_t *out,
                               uint32_t bit) {
  if (bit > 32 || length == 0) return in;
  const uint8_t *p = static_cast<const uint8_t *>(in);
  Call call;
  Staging &s = *call.ctx;
  const size_t total = simdcomp_hip_compressedbytes(length, bit);
  if (length * 4 <= ZC_BYTES) {
    hipStream_t st = s.stream[0];
    memcpy(s.zi(), p, total);
    uint32_t *z_out = reinterpret_cast<uint32_t *>(s.zout);
    if (kind == 0) ckapi(simdcomp_hip_unpack_length(s.zi(), length, z_out, bit, st), "unpack_length");
    else ckapi(simdcomp_hip_unpackFOR_length(initvalue, s.zi(), length, z_out, bit, st), "unpackFOR_length");
    s.finish(st);
    memcpy(out, z_out, length * 4);
    return p + total;
  }
  if (void *din = bit ? device_view(in) : const_cast<void *>(in)) {  // (a 0-bit array has no payload to look at)
    if (void *dout = device_view(out)) {
      hipStream_t st = s.stream[0];
      uint32_t *d_out = static_cast<uint32_t *>(dout);
      if (kind == 0) ckapi(simdcomp_hip_unpack_length(din, length, d_out, bit, st), "unpack_length");
      else ckapi(simdcomp_hip_unpackFOR_length(initvalue, din, length, d_out, bit, st), "unpackFOR_length");
      ck(hipStreamSynchronize(st), "hipStreamSynchronize");
      return p + total;
    }
  }
  run_pipeline(
      s, length, simdcomp_hip_compressedbytes(PIECE, bit), PIECE * 4,
      [&](int q, size_t done, size_t n, hipStream_t st) {
        h2d(s.din(q), p + (done / 128) * 16 * bit, simdcomp_hip_compressedbytes(n, bit), st);
      },
      [&](int q, size_t, size_t n, hipStream_t st) {
        uint32_t *d_out = reinterpret_cast<uint32_t *>(s.dout(q));
        if (kind == 0) ckapi(simdcomp_hip_unpack_length(s.din(q), n, d_out, bit, st), "unpack_length");
        else ckapi(simdcomp_hip_unpackFOR_length(initvalue, s.din(q), n, d_out, bit, st), "unpackFOR_length");
      },
      [&](int q, size_t done, size_t n, hipStream_t st) { d2h(out + done, s.dout(q), n * 4, st); });
  return p + total;
}

}  // namespace

extern "C" {

void simdcomp_hip_set_error_handler(simdcomp_hip_error_handler handler, void *arg) {
  g_handler_arg.store(arg);
  g_handler.store(handler);
}

// ---- include/simdcomputil.h of the reference ------------------------------------
uint32_t maxbits(const uint32_t *begin) { return maxbits_length(begin, SIMDBlockSize); }

uint32_t maxbits_length(const uint32_t *in, uint32_t length) {
  uint32_t r[4];
  reduce_host(in, length, 0, r);
  return width_of(r[0]);
}

uint32_t simdmin(const uint32_t *in) { return simdmin_length(in, SIMDBlockSize); }

uint32_t simdmin_length(const uint32_t *in, uint32_t length) {
  uint32_t r[4];
  reduce_host(in, length, 0, r);
  return r[2];
}

void simdmaxmin_length(const uint32_t *in, uint32_t length, uint32_t *getmin, uint32_t *getmax) {
  uint32_t r[4];
  reduce_host(in, length, 0, r);
  *getmin = r[2];
  *getmax = r[3];
}

void simdmaxmin(const uint32_t *in, uint32_t *getmin, uint32_t *getmax) {
  simdmaxmin_length(in, SIMDBlockSize, getmin, getmax);
}

uint32_t simdmaxbitsd1(uint32_t initvalue, const uint32_t *in) {
  return simdmaxbitsd1_length(initvalue, in, SIMDBlockSize);
}

uint32_t simdmaxbitsd1_length(uint32_t initvalue, const uint32_t *in, uint32_t length) {
  uint32_t r[4];
  reduce_host(in, length, initvalue, r);
  return width_of(r[1]);
}

// ---- include/simdbitpacking.h ------------------------------------------------------
void simdpack(const uint32_t *in, __m128i *out, const uint32_t bit) {
  block_pack(sc::MODE_PLAIN, true, 0, in, out, bit);
}

void simdpackwithoutmask(const uint32_t *in, __m128i *out, const uint32_t bit) {
  block_pack(sc::MODE_PLAIN, false, 0, in, out, bit);
}

void simdunpack(const __m128i *in, uint32_t *out, const uint32_t bit) {
  block_unpack(sc::MODE_PLAIN, 0, in, out, bit);
}

// ---- the 8- and 16-lane wire formats (include/avxbitpacking.h:22-31, include/avx512bitpacking.h) ----
// Declared with __m256i / __m512i in simdcomp.h when the including program is built for those ISAs; the library
// itself only moves bytes, so the definitions take untyped pointers.
uint32_t avxmaxbits(const uint32_t *begin) { return maxbits_length(begin, 256); }
void avxpack(const uint32_t *in, void *out, const uint32_t bit) { block_pack(sc::MODE_PLAIN, true, 0, in, out, bit, 2); }
void avxpackwithoutmask(const uint32_t *in, void *out, const uint32_t bit) { block_pack(sc::MODE_PLAIN, false, 0, in, out, bit, 2); }
void avxunpack(const void *in, uint32_t *out, const uint32_t bit) { block_unpack(sc::MODE_PLAIN, 0, in, out, bit, 2); }
uint32_t avx512maxbits(const uint32_t *begin) { return maxbits_length(begin, 512); }
void avx512pack(const uint32_t *in, void *out, const uint32_t bit) { block_pack(sc::MODE_PLAIN, true, 0, in, out, bit, 4); }
void avx512packwithoutmask(const uint32_t *in, void *out, const uint32_t bit) { block_pack(sc::MODE_PLAIN, false, 0, in, out, bit, 4); }
void avx512unpack(const void *in, uint32_t *out, const uint32_t bit) { block_unpack(sc::MODE_PLAIN, 0, in, out, bit, 4); }

__m128i *simdpack_length(const uint32_t *in, size_t length, __m128i *out, const uint32_t bit) {
  return static_cast<__m128i *>(pack_length_host(0, 0, in, length, out, bit));
}

const __m128i *simdunpack_length(const __m128i *in, size_t length, uint32_t *out, const uint32_t bit) {
  return static_cast<const __m128i *>(unpack_length_host(0, 0, in, length, out, bit));
}

// The short packer is the unmasked arbitrary-length loop on its own; the frame
// kernel with base 0 is exactly that loop (src/simdbitpacking.c:14190 and
// src/simdfor.c:15043 differ only in the subtraction and the padding value).
__m128i *simdpack_shortlength(const uint32_t *in, int length, __m128i *out, const uint32_t bit) {
  return static_cast<__m128i *>(pack_length_host(1, 0, in, (size_t)(length < 0 ? 0 : length), out, bit));
}

const __m128i *simdunpack_shortlength(const __m128i *in, int length, uint32_t *out, const uint32_t bit) {
  return static_cast<const __m128i *>(unpack_length_host(0, 0, in, (size_t)(length < 0 ? 0 : length), out, bit));
}

// ---- include/simdintegratedbitpacking.h ------------------------------------------------
void simdpackd1(uint32_t initvalue, const uint32_t *in, __m128i *out, const uint32_t bit) {
  block_pack(sc::MODE_D1, true, initvalue, in, out, bit);
}

void simdpackwithoutmaskd1(uint32_t initvalue, const uint32_t *in, __m128i *out, const uint32_t bit) {
  block_pack(sc::MODE_D1, false, initvalue, in, out, bit);
}

void simdunpackd1(uint32_t initvalue, const __m128i *in, uint32_t *out, const uint32_t bit) {
  block_unpack(sc::MODE_D1, initvalue, in, out, bit);
}

// ---- include/simdfor.h ---------------------------------------------------------------------
void simdpackFOR(uint32_t initvalue, const uint32_t *in, __m128i *out, const uint32_t bit) {
  block_pack(sc::MODE_FOR, false, initvalue, in, out, bit);
}

void simdunpackFOR(uint32_t initvalue, const __m128i *in, uint32_t *out, const uint32_t bit) {
  block_unpack(sc::MODE_FOR, initvalue, in, out, bit);
}

__m128i *simdpackFOR_length(uint32_t initvalue, const uint32_t *in, int length, __m128i *out, const uint32_t bit) {
  return static_cast<__m128i *>(pack_length_host(1, initvalue, in, (size_t)(length < 0 ? 0 : length), out, bit));
}

const __m128i *simdunpackFOR_length(uint32_t initvalue, const __m128i *in, int length, uint32_t *out,
                                    const uint32_t bit) {
  return static_cast<const __m128i *>(
      unpack_length_host(1, initvalue, in, (size_t)(length < 0 ? 0 : length), out, bit));
}

}  // extern "C"
