"""The reference-signature host API (include/simdcomp.h) of libsimdcomp_hip.so,
called through ctypes with HOST arrays exactly as a C caller of the reference
would, compared with the golden fixtures and the oracle.  The tests read like
the reference's own (tests/unit.c): pack, check the returned pointer against
*_compressedbytes, unpack, compare."""
import ctypes as C

import numpy as np
import pytest

import golden_util as G
from simdcomp_amd import binding, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return binding.lib()


def p(a):
    return a.ctypes.data


def test_blocks_against_golden(lib):
    g = G.load("blocks.npz")
    for bit in range(33):
        for kind in range(3):
            x = g["x"][bit, kind].copy()
            iv = int(g["init"][bit, kind])
            outs = []
            for fn, seeded in ((lib.simdpack, False), (lib.simdpackwithoutmask, False), (lib.simdpackd1, True),
                               (lib.simdpackwithoutmaskd1, True), (lib.simdpackFOR, True)):
                buf = np.full(512 + 16, 0xA5, np.uint8)
                if seeded:
                    fn(iv, p(x), p(buf), bit)
                else:
                    fn(p(x), p(buf), bit)
                assert (buf[16 * bit:] == 0xA5).all(), "packer wrote past b vectors"
                outs.append(buf[:16 * bit].copy())
            for j, o in enumerate(outs):
                want = g["packed"][bit, kind, j, :16 * bit]
                assert np.array_equal(o, want), (G.PACK_NAMES[j], bit, kind)
                src = np.concatenate([want, np.zeros(16, np.uint8)])
                out = np.zeros(128, np.uint32)
                lib.simdunpack(p(src), p(out), bit)
                assert np.array_equal(out, g["unpacked"][bit, kind, j, 0])
                lib.simdunpackd1(iv, p(src), p(out), bit)
                assert np.array_equal(out, g["unpacked"][bit, kind, j, 1])
                lib.simdunpackFOR(iv, p(src), p(out), bit)
                assert np.array_equal(out, g["unpacked"][bit, kind, j, 2])
            lo, hi = C.c_uint32(), C.c_uint32()
            lib.simdmaxmin(p(x), C.byref(lo), C.byref(hi))
            red = [lib.maxbits(p(x)), lib.simdmaxbitsd1(iv, p(x)), lo.value, hi.value]
            assert red == g["red"][bit, kind].tolist()
            assert lib.simdmin(p(x)) == lo.value


def test_bit_above_32_does_nothing(lib):
    x = synth.rand_u32(1, 128)
    buf = np.full(600, 0xA5, np.uint8)
    lib.simdpack(p(x), p(buf), 33)
    lib.simdpackd1(7, p(x), p(buf), 40)
    assert (buf == 0xA5).all()
    out = np.full(128, 0xDEADBEEF, np.uint32)
    lib.simdunpack(p(buf), p(out), 33)
    assert (out == 0xDEADBEEF).all()


def test_unaligned_pointers_like_example_c(lib, orc):
    """example.c:68-72 feeds payloads shifted by the 1-byte width header."""
    x = synth.sorted_gaps(5, 128, 11, first=100)
    raw = np.zeros(1 + 16 * 11 + 8, np.uint8)
    b = lib.simdmaxbitsd1(100, p(x))
    assert b == orc.simdmaxbitsd1(100, x)
    raw[0] = b
    lib.simdpackwithoutmaskd1(100, p(x), p(raw) + 1, b)
    assert np.array_equal(raw[1:1 + 16 * b], orc.simdpackwithoutmaskd1(100, x, b))
    xs = np.zeros(129, np.uint32)[1:]  # 4-byte aligned, not 16
    out = np.zeros(129, np.uint32)[1:]
    lib.simdunpackd1(100, p(raw) + 1, p(out), int(raw[0]))
    assert np.array_equal(out, x)


def test_example_c_stream_round_trip(lib, orc):
    """compress() / decode loop of example.c:56-75,101-110 through the host API."""
    nblk = 20
    x = synth.sorted_gaps(6, nblk * 128, 9, first=0)
    stream = np.zeros(nblk * (1 + 512), np.uint8)
    pos, offset = 0, 0
    for k in range(nblk):
        blk = x[k * 128:(k + 1) * 128].copy()
        b = lib.simdmaxbitsd1(offset, p(blk))
        stream[pos] = b
        pos += 1
        lib.simdpackwithoutmaskd1(offset, p(blk), p(stream) + pos, b)
        offset = int(blk[127])
        pos += 16 * b
    back = np.zeros(128, np.uint32)
    rp, offset = 0, 0
    for k in range(nblk):
        b = int(stream[rp])
        rp += 1
        lib.simdunpackd1(offset, p(stream) + rp, p(back), b)
        assert np.array_equal(back, x[k * 128:(k + 1) * 128])
        rp += 16 * b
        offset = int(back[127])
    assert rp == pos


@pytest.mark.parametrize("bit", [0, 1, 3, 7, 8, 13, 16, 21, 31, 32])
def test_lengths(lib, orc, bit):
    """issue21 / testshortpack / testlongpack / testshortFORpack of tests/unit.c, with oracle comparison."""
    data = {"inrange": synth.random_bits(7, 3000, bit), "wild": synth.rand_u32(8, 3000)}
    for name, full in data.items():
        for n in [0, 1, 2, 3, 4, 5, 31, 127, 128, 129, 255, 257, 1000, 2047, 3000]:
            x = full[:n].copy() if n else np.zeros(4, np.uint32)[:0]
            xs = np.concatenate([x, np.zeros(4, np.uint32)])
            nbytes = lib.simdpack_compressedbytes(n, bit)
            buf = np.full(nbytes + 64, 0xA5, np.uint8)
            end = lib.simdpack_length(p(xs), n, p(buf), bit)
            assert end - p(buf) == nbytes, "returned pointer != compressedbytes"
            assert np.array_equal(buf[:nbytes], orc.simdpack_length(x, bit)), ("simdpack_length", name, n, bit)
            assert (buf[nbytes:] == 0xA5).all()
            out = np.full(n + 8, 0xDEADBEEF, np.uint32)
            rend = lib.simdunpack_length(p(buf), n, p(out), bit)
            assert rend - p(buf) == (nbytes if n else 0)
            want, _ = orc.simdunpack_length(buf[:nbytes], n, bit)
            assert np.array_equal(out[:n], want) and (out[n:] == 0xDEADBEEF).all()
            if name == "inrange":
                assert np.array_equal(out[:n], x)
            base = 7 if name == "inrange" else 0x9E3779B9
            xf = (xs + np.uint32(base)).astype(np.uint32) if name == "inrange" else xs
            buf = np.full(nbytes + 64, 0xA5, np.uint8)
            end = lib.simdpackFOR_length(base, p(xf), n, p(buf), bit)
            assert end - p(buf) == lib.simdpackFOR_compressedbytes(n, bit) == nbytes
            assert np.array_equal(buf[:nbytes], orc.simdpackFOR_length(base, xf[:n].copy(), bit)), \
                ("simdpackFOR_length", name, n, bit)
            assert (buf[nbytes:] == 0xA5).all()
            out = np.full(n + 8, 0xDEADBEEF, np.uint32)
            lib.simdunpackFOR_length(base, p(buf), n, p(out), bit)
            want, _ = orc.simdunpackFOR_length(base, buf[:nbytes], n, bit)
            assert np.array_equal(out[:n], want) and (out[n:] == 0xDEADBEEF).all()
            if n <= 128:
                buf = np.full(nbytes + 64, 0xA5, np.uint8)
                end = lib.simdpack_shortlength(p(xs), n, p(buf), bit)
                assert end - p(buf) == nbytes
                assert np.array_equal(buf[:nbytes], orc.simdpack_shortlength(x, bit)), ("short", name, n, bit)
                out = np.full(n + 8, 0xDEADBEEF, np.uint32)
                lib.simdunpack_shortlength(p(buf), n, p(out), bit)
                want, _ = orc.simdunpack_shortlength(buf[:nbytes], n, bit)
                assert np.array_equal(out[:n], want) and (out[n:] == 0xDEADBEEF).all()
            assert lib.maxbits_length(p(xs), n) == orc.maxbits_length(xs, n)
            assert lib.simdmin_length(p(xs), n) == orc.simdmin_length(xs, n)
            lo, hi = C.c_uint32(), C.c_uint32()
            lib.simdmaxmin_length(p(xs), n, C.byref(lo), C.byref(hi))
            assert (lo.value, hi.value) == orc.simdmaxmin_length(xs, n)
            if n:
                assert lib.simdmaxbitsd1_length(9, p(xs), n) == orc.simdmaxbitsd1_length(9, xs, n)


def test_long_array_crosses_pipeline_pieces(lib, orc):
    """More 2^24-integer pieces than the pipeline has staging slots (so slots are reused while uploads, kernels and
    downloads of neighbouring pieces are in flight), ragged end, arbitrary 32-bit data (so the frame packer's
    hand-over between blocks is exercised across the piece cuts)."""
    n = 3 * (1 << 24) + (1 << 18) + 77
    x = synth.rand_u32(9, n)
    for bit in (5, 32):
        nbytes = lib.simdpack_compressedbytes(n, bit)
        buf = np.zeros(nbytes + 16, np.uint8)
        end = lib.simdpackFOR_length(12345, p(x), n, p(buf), bit)
        assert end - p(buf) == nbytes
        want = orc.simdpackFOR_length(12345, x, bit)
        assert orc.fnv1a64(buf[:nbytes]) == orc.fnv1a64(want)
        out = np.zeros(n, np.uint32)
        lib.simdunpackFOR_length(12345, p(buf), n, p(out), bit)
        wdec, _ = orc.simdunpackFOR_length(12345, want, n, bit)
        assert orc.fnv1a64(out) == orc.fnv1a64(wdec)
        end = lib.simdpack_length(p(x), n, p(buf), bit)
        assert end - p(buf) == nbytes
        assert orc.fnv1a64(buf[:nbytes]) == orc.fnv1a64(orc.simdpack_length(x, bit))
    assert lib.maxbits_length(p(x), n) == 32
    assert lib.simdmaxbitsd1_length(0, p(x), n) == 32


def test_long_arrays_in_pinned_memory_are_coded_in_place(lib, orc):
    """Caller buffers in pinned host memory take the path without staging (the kernels read and write them over the
    host link): same bytes as the staged path, nothing written past the end; a pinned buffer at an address that is not
    16-byte aligned, or only one pinned side, goes through staging and gives the same bytes again."""
    import torch

    n = (1 << 22) + 77
    x = synth.rand_u32(19, n)
    hx = torch.empty(n + 4, dtype=torch.int32, pin_memory=True)
    hx[:n] = torch.from_numpy(x.view(np.int32))
    for bit in (7, 32):
        nbytes = lib.simdpack_compressedbytes(n, bit)
        want = orc.simdpackFOR_length(777, x, bit)
        wdec, _ = orc.simdunpackFOR_length(777, want, n, bit)
        hbuf = torch.full((nbytes + 64,), 0xA5, dtype=torch.uint8, pin_memory=True)
        hout = torch.full((n + 8,), 0x5A5A5A5A, dtype=torch.int32, pin_memory=True)
        for shift in (0, 4):  # 0: in place; 4: the payload is not 16-byte aligned -> staged
            hbuf.fill_(0xA5)
            end = lib.simdpackFOR_length(777, hx.data_ptr(), n, hbuf.data_ptr() + shift, bit)
            assert end - (hbuf.data_ptr() + shift) == nbytes
            got = hbuf.numpy()
            assert np.array_equal(got[shift:shift + nbytes], want), (bit, shift)
            assert (got[:shift] == 0xA5).all() and (got[shift + nbytes:] == 0xA5).all(), "wrote outside the payload"
            hout.fill_(0x5A5A5A5A)
            lib.simdunpackFOR_length(777, hbuf.data_ptr() + shift, n, hout.data_ptr(), bit)
            assert np.array_equal(hout.numpy()[:n].view(np.uint32), wdec), (bit, shift)
            assert (hout[n:] == 0x5A5A5A5A).all()
        # plain coding, pinned input and pageable output
        buf = np.zeros(nbytes + 16, np.uint8)
        end = lib.simdpack_length(hx.data_ptr(), n, p(buf), bit)
        assert end - p(buf) == nbytes and np.array_equal(buf[:nbytes], orc.simdpack_length(x, bit))
        out = np.zeros(n, np.uint32)
        hbuf[:nbytes] = torch.from_numpy(buf[:nbytes])
        lib.simdunpack_length(hbuf.data_ptr(), n, p(out), bit)
        wdec = orc.simdunpack_length(buf[:nbytes], n, bit)
        wdec = wdec[0] if isinstance(wdec, tuple) else wdec
        assert np.array_equal(out, wdec)  # (whole blocks are masked, the 77 values behind them are not)
        lib.simdunpack_length(hbuf.data_ptr(), n, hout.data_ptr(), bit)
        assert np.array_equal(hout.numpy()[:n].view(np.uint32), out)


def test_device_checksum_matches_host_formula():
    import torch

    from simdcomp_amd import codec

    x = synth.rand_u32(70, 100000)
    idx = (np.arange(len(x), dtype=np.uint64) + np.uint64(12345)) << np.uint64(32)
    want = int(synth.splitmix64(idx ^ x.astype(np.uint64)).sum(dtype=np.uint64))
    acc = codec.checksum(torch.from_numpy(x.view(np.int32)).cuda(), index0=12345)
    assert int(acc[0].item()) & (2 ** 64 - 1) == want


def test_concurrent_callers_on_disjoint_buffers(lib, orc):
    """The reference is stateless (src/simdbitpacking.c:14379-14399 touch nothing but their arguments), so threads may
    call it at the same time on disjoint buffers.  Four host threads do that here -- block calls, short and long
    *_length calls, reductions -- and every result is compared with the oracle.  (ctypes drops the GIL during a
    foreign call, so the calls really overlap.)"""
    import threading

    nthreads, errors = 4, []
    start = threading.Barrier(nthreads)

    def worker(t):
        try:
            rng_seed = 100 + t
            n_long = (1 << 22) + 77 * (t + 1)  # several 64 KiB+ pieces through the staged path
            bit = 5 + 6 * t
            x_long = synth.random_bits(rng_seed, n_long, bit)
            want_long = orc.simdpack_length(x_long, bit)
            blocks = [synth.sorted_gaps(rng_seed + i, 128, min(bit, 20), first=t) for i in range(40)]
            want_blocks = [orc.simdpackwithoutmaskd1(t, b, bit) for b in blocks]
            start.wait()
            for rep in range(3):
                out = np.full(want_long.size + 64, 0xA5, np.uint8)
                end = lib.simdpack_length(p(x_long), n_long, p(out), bit)
                assert end - p(out) == want_long.size
                assert np.array_equal(out[:want_long.size], want_long) and (out[want_long.size:] == 0xA5).all()
                back = np.zeros(n_long, np.uint32)
                lib.simdunpack_length(p(out), n_long, p(back), bit)
                assert np.array_equal(back, x_long)
                for b, w in zip(blocks, want_blocks):
                    buf = np.zeros(16 * bit + 16, np.uint8)
                    lib.simdpackwithoutmaskd1(t, p(b), p(buf), bit)
                    assert np.array_equal(buf[:16 * bit], w)
                    dec = np.zeros(128, np.uint32)
                    lib.simdunpackd1(t, p(buf), p(dec), bit)
                    assert np.array_equal(dec, b)
                    assert lib.simdmaxbitsd1(t, p(b)) == orc.simdmaxbitsd1(t, b)
                assert lib.maxbits_length(p(x_long), 100000) == orc.maxbits_length(x_long, 100000)
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))
            try:
                start.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=280)
        assert not th.is_alive(), "a caller thread hung"
    assert not errors, errors


def test_host_api_follows_the_callers_current_device(lib, orc):
    """A host call runs on the HIP device that is current in the calling thread (on a one-GPU box: device 0, before
    and after), and leaves the caller's device selection as it found it."""
    import torch

    before = torch.cuda.current_device()
    x = synth.random_bits(3, 128, 9)
    buf = np.zeros(16 * 9, np.uint8)
    lib.simdpack(p(x), p(buf), 9)
    assert np.array_equal(buf, orc.simdpack(x, 9))
    assert torch.cuda.current_device() == before


def test_small_calls_with_and_without_polled_completion():
    """Small calls wait for a word the stream writes behind the kernel; SIMDCOMP_HIP_POLL=0 goes through
    hipStreamSynchronize instead.  Both give the reference's bytes (a fresh process each: the switch is read once)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = r"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, %r)
import oracle
from simdcomp_amd import binding
L = binding.lib()
vp = C.c_void_p
L.simdpackwithoutmaskd1.argtypes = [C.c_uint32, vp, vp, C.c_uint32]; L.simdpackwithoutmaskd1.restype = None
L.simdunpackd1.argtypes = [C.c_uint32, vp, vp, C.c_uint32]; L.simdunpackd1.restype = None
L.simdpack_length.argtypes = [vp, C.c_size_t, vp, C.c_uint32]; L.simdpack_length.restype = vp
rng = np.random.default_rng(5)
orc = oracle.Oracle()
for it in range(300):
    b = int(rng.integers(0, 33))
    x = np.cumsum(rng.integers(0, 1 << min(b, 24), 128, dtype=np.uint64)).astype(np.uint32)
    out = np.zeros(4 * max(b, 1), np.uint32); back = np.zeros(128, np.uint32)
    L.simdpackwithoutmaskd1(0, x.ctypes.data, out.ctypes.data, b)
    assert np.array_equal(out.view(np.uint8)[:16 * b], orc.simdpackwithoutmaskd1(0, x, b)), (it, b)
    L.simdunpackd1(0, out.ctypes.data, back.ctypes.data, b)
    assert np.array_equal(back, orc.simdunpackd1(0, out, b)) and np.array_equal(back, x), (it, b)
n = 1000
y = rng.integers(0, 1 << 11, n, dtype=np.uint32)
buf = np.zeros(n + 64, np.uint32)
end = L.simdpack_length(y.ctypes.data, n, buf.ctypes.data, 11)
assert end - buf.ctypes.data == orc.compressedbytes(n, 11)
print("ok")
""" % root
    for poll in ("1", "0"):
        env = dict(os.environ, SIMDCOMP_HIP_POLL=poll)
        out = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (poll, out.stdout[-500:], out.stderr[-2000:])


def test_an_earlier_hip_error_of_the_caller_is_not_reported_as_ours(lib, orc):
    """HIP keeps a thread's last error until somebody reads it.  A failed HIP call of the application (here: an
    unregister of memory that was never registered) must not come back as the result of the next library call,
    batched or through the reference signatures."""
    import torch

    from simdcomp_amd import codec

    rt = torch.cuda.cudart()
    x = synth.random_bits(5, 128 * 40, 9)
    d_x = torch.from_numpy(x.view(np.int32)).cuda()
    scratch = np.zeros(4096, np.uint8)
    for _ in range(2):
        try:
            rc = rt.cudaHostUnregister(scratch.ctypes.data)  # fails: sets the thread's last error
        except RuntimeError:
            rc = 1
        assert rc != 0 and int(rc) != 0
        pay = codec.pack(d_x, 9, masked=True)  # raises if the call reports an error
        assert np.array_equal(pay.cpu().numpy(), orc.batch_pack("plain", x, 9, masked=True))
        try:
            rt.cudaHostUnregister(scratch.ctypes.data)
        except RuntimeError:
            pass
        out = np.zeros(128, np.uint32)
        buf = np.zeros(16 * 9, np.uint8)
        lib.simdpack(p(x[:128].copy()), p(buf), 9)
        lib.simdunpack(p(buf), p(out), 9)
        assert np.array_equal(out, x[:128])
