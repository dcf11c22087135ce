"""TEST INFRASTRUCTURE ONLY: ctypes bindings for the CPU checkers.

Two libraries expose the same block codec under two symbol prefixes:

* ``oracle/liboracle.so`` -- the scalar restatement (``orc_`` prefix) plus the
  batched drivers and the CPU timing driver;
* ``oracle/_ref/libsimdcomp_ref.so`` -- the reference itself, compiled from its
  own sources by ``oracle/Makefile`` (no prefix).  Optional: present only when
  it was built in the container that has the reference tree.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg import this package.  The shipped library never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libsimdcomp_ref.so")

u32 = C.c_uint32
vp = C.c_void_p


def build(verbose=False):
    """Compile liboracle.so (and _ref when the reference tree is present)."""
    out = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout + out.stderr)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed")


def _ptr(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(vp)


class BlockCodec:
    """Uniform view of one implementation of the reference's block API.

    ``prefix`` is ``"orc_"`` for the restatement and ``""`` for the compiled
    reference.  All arrays are numpy; compressed data is ``uint8``.
    """

    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.path = path
        L, p = self.lib, prefix

        def fn(name, res, args):
            f = getattr(L, p + name)
            f.restype = res
            f.argtypes = args
            return f

        self._bits = fn("bits", u32, [u32])
        self._maxbits = fn("maxbits", u32, [vp])
        self._maxbits_length = fn("maxbits_length", u32, [vp, u32])
        self._maxbitsd1 = fn("simdmaxbitsd1", u32, [u32, vp])
        self._maxbitsd1_length = fn("simdmaxbitsd1_length", u32, [u32, vp, u32])
        self._min = fn("simdmin", u32, [vp])
        self._min_length = fn("simdmin_length", u32, [vp, u32])
        self._maxmin = fn("simdmaxmin", None, [vp, vp, vp])
        self._maxmin_length = fn("simdmaxmin_length", None, [vp, u32, vp, vp])
        self._pack = fn("simdpack", None, [vp, vp, u32])
        self._packwm = fn("simdpackwithoutmask", None, [vp, vp, u32])
        self._unpack = fn("simdunpack", None, [vp, vp, u32])
        self._packd1 = fn("simdpackd1", None, [u32, vp, vp, u32])
        self._packwmd1 = fn("simdpackwithoutmaskd1", None, [u32, vp, vp, u32])
        self._unpackd1 = fn("simdunpackd1", None, [u32, vp, vp, u32])
        self._packFOR = fn("simdpackFOR", None, [u32, vp, vp, u32])
        self._unpackFOR = fn("simdunpackFOR", None, [u32, vp, vp, u32])
        self._cbytes = fn("simdpack_compressedbytes", C.c_int, [C.c_int, u32])
        self._cbytesFOR = fn("simdpackFOR_compressedbytes", C.c_int, [C.c_int, u32])
        self._pack_short = fn("simdpack_shortlength", vp, [vp, C.c_int, vp, u32])
        self._unpack_short = fn("simdunpack_shortlength", vp, [vp, C.c_int, vp, u32])
        self._pack_length = fn("simdpack_length", vp, [vp, C.c_size_t, vp, u32])
        self._unpack_length = fn("simdunpack_length", vp, [vp, C.c_size_t, vp, u32])
        self._packFOR_length = fn("simdpackFOR_length", vp, [u32, vp, C.c_int, vp, u32])
        self._unpackFOR_length = fn("simdunpackFOR_length", vp, [u32, vp, C.c_int, vp, u32])
        self._selectFOR = fn("simdselectFOR", u32, [u32, vp, u32, C.c_int])
        self._searchFOR = fn("simdsearchwithlengthFOR", C.c_int, [u32, vp, u32, C.c_int, u32, vp])
        self._selectd1 = fn("simdselectd1", u32, [u32, vp, u32, C.c_int])
        self._searchlend1 = fn("simdsearchwithlengthd1", C.c_int, [u32, vp, u32, C.c_int, u32, vp])
        self._searchd1 = fn("simdsearchd1", C.c_int, [vp, vp, u32, u32, vp])
        self._scand1 = fn("simdscand1", None, [vp, vp, u32])
        self._fastset = fn("simdfastset", None, [vp, u32, u32, C.c_size_t])
        self._fastsetFOR = fn("simdfastsetFOR", None, [u32, vp, u32, u32, C.c_size_t])

    # -- reductions ---------------------------------------------------------
    def bits(self, v):
        return int(self._bits(int(v)))

    def maxbits(self, x):
        return int(self._maxbits(_ptr(x)))

    def maxbits_length(self, x, n=None):
        return int(self._maxbits_length(_ptr(x), len(x) if n is None else n))

    def simdmaxbitsd1(self, init, x):
        return int(self._maxbitsd1(int(init), _ptr(x)))

    def simdmaxbitsd1_length(self, init, x, n=None):
        return int(self._maxbitsd1_length(int(init), _ptr(x), len(x) if n is None else n))

    def simdmin(self, x):
        return int(self._min(_ptr(x)))

    def simdmin_length(self, x, n=None):
        return int(self._min_length(_ptr(x), len(x) if n is None else n))

    def simdmaxmin(self, x):
        lo, hi = u32(), u32()
        self._maxmin(_ptr(x), C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def simdmaxmin_length(self, x, n=None):
        lo, hi = u32(), u32()
        self._maxmin_length(_ptr(x), len(x) if n is None else n, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    # -- one block ----------------------------------------------------------
    @staticmethod
    def _cbuf(bit, fill=0):
        return np.full(16 * min(int(bit), 32), fill, dtype=np.uint8)

    def simdpack(self, x, bit):
        o = self._cbuf(bit)
        self._pack(_ptr(x), _ptr(o), bit)
        return o

    def simdpackwithoutmask(self, x, bit):
        o = self._cbuf(bit)
        self._packwm(_ptr(x), _ptr(o), bit)
        return o

    def simdunpack(self, comp, bit):
        o = np.zeros(128, dtype=np.uint32)
        self._unpack(_ptr(comp), _ptr(o), bit)
        return o

    def simdpackd1(self, init, x, bit):
        o = self._cbuf(bit)
        self._packd1(int(init), _ptr(x), _ptr(o), bit)
        return o

    def simdpackwithoutmaskd1(self, init, x, bit):
        o = self._cbuf(bit)
        self._packwmd1(int(init), _ptr(x), _ptr(o), bit)
        return o

    def simdunpackd1(self, init, comp, bit):
        o = np.zeros(128, dtype=np.uint32)
        self._unpackd1(int(init), _ptr(comp), _ptr(o), bit)
        return o

    def simdpackFOR(self, base, x, bit):
        # the reference's FOR block kernels use aligned 16-byte accesses
        xa = aligned_copy(x)
        o = aligned_empty(16 * bit, np.uint8)
        o[:] = 0
        self._packFOR(int(base), _ptr(xa), _ptr(o), bit)
        return o.copy()

    def simdunpackFOR(self, base, comp, bit):
        ca = aligned_copy(comp)
        o = aligned_empty(128, np.uint32)
        o[:] = 0
        self._unpackFOR(int(base), _ptr(ca), _ptr(o), bit)
        return o.copy()

    # -- arbitrary length ---------------------------------------------------
    def compressedbytes(self, n, bit):
        return int(self._cbytes(int(n), bit))

    def compressedbytesFOR(self, n, bit):
        return int(self._cbytesFOR(int(n), bit))

    def _run_pack_len(self, f, args_before, x, n, bit, cap):
        o = np.zeros(cap + 64, dtype=np.uint8)
        o[:] = 0xA5  # canary: bytes past the returned pointer must stay untouched
        end = f(*args_before, _ptr(x) if len(x) else _ptr(np.zeros(4, np.uint32)), n, _ptr(o), bit)
        used = int(end) - o.ctypes.data
        assert 0 <= used <= cap + 64
        assert (o[used:] == 0xA5).all(), "packer wrote past its returned pointer"
        return o[:used].copy()

    def simdpack_length(self, x, bit):
        n = len(x)
        return self._run_pack_len(self._pack_length, (), x, n, bit, 4 * n + 64)

    def simdpack_shortlength(self, x, bit):
        n = len(x)
        return self._run_pack_len(self._pack_short, (), x, n, bit, 4 * n + 64)

    def simdpackFOR_length(self, base, x, bit):
        n = len(x)
        return self._run_pack_len(self._packFOR_length, (int(base),), x, n, bit, 4 * n + 64)

    def _run_unpack_len(self, f, args_before, comp, n, bit):
        o = np.zeros(n + 8, dtype=np.uint32)
        o[:] = 0xDEADBEEF
        c = np.concatenate([np.asarray(comp, np.uint8), np.zeros(64, np.uint8)])
        end = f(*args_before, _ptr(c), n, _ptr(o), bit)
        used = int(end) - c.ctypes.data
        assert (o[n:] == 0xDEADBEEF).all(), "unpacker wrote past length"
        return o[:n].copy(), used

    def simdunpack_length(self, comp, n, bit):
        return self._run_unpack_len(self._unpack_length, (), comp, n, bit)

    def simdunpack_shortlength(self, comp, n, bit):
        return self._run_unpack_len(self._unpack_short, (), comp, n, bit)

    def simdunpackFOR_length(self, base, comp, n, bit):
        return self._run_unpack_len(self._unpackFOR_length, (int(base),), comp, n, bit)

    # -- single values ------------------------------------------------------
    def simdselectFOR(self, base, comp, bit, slot):
        c = np.concatenate([np.asarray(comp, np.uint8), np.zeros(32, np.uint8)])
        return int(self._selectFOR(int(base), _ptr(c), bit, slot))

    def simdsearchwithlengthFOR(self, base, comp, bit, length, key):
        c = np.concatenate([np.asarray(comp, np.uint8), np.zeros(32, np.uint8)])
        r = u32()
        pos = self._searchFOR(int(base), _ptr(c), bit, length, int(key), C.byref(r))
        return int(pos), r.value

    @staticmethod
    def _block512(comp):
        """The payload in a 16-byte aligned 512-byte buffer (the search routines read whole vectors of a 32-bit block)."""
        c = aligned_empty(512 + 64, np.uint8)
        c[:] = 0
        comp = np.asarray(comp, np.uint8)
        c[:comp.size] = comp
        return c

    def simdselectd1(self, init, comp, bit, slot):
        return int(self._selectd1(int(init), _ptr(self._block512(comp)), bit, slot))

    def simdsearchwithlengthd1(self, init, comp, bit, length, key):
        r = u32()
        pos = self._searchlend1(int(init), _ptr(self._block512(comp)), bit, length, int(key), C.byref(r))
        return int(pos), r.value

    def simdsearchd1(self, offset4, comp, bit, key):
        """-> (position, value, offset vector after the call)"""
        o = aligned_copy(np.asarray(offset4, np.uint32))
        r = u32()
        pos = self._searchd1(_ptr(o), _ptr(self._block512(comp)), bit, int(key), C.byref(r))
        return int(pos), r.value, o.copy()

    def simdscand1(self, offset4, comp, bit):
        o = aligned_copy(np.asarray(offset4, np.uint32))
        self._scand1(_ptr(o), _ptr(self._block512(comp)), bit)
        return o.copy()

    def simdfastset(self, comp, bit, value, index):
        c = np.asarray(comp, np.uint8).copy()
        self._fastset(_ptr(c), bit, int(value), index)
        return c

    def simdfastsetFOR(self, base, comp, bit, value, index):
        c = np.asarray(comp, np.uint8).copy()
        self._fastsetFOR(int(base), _ptr(c), bit, int(value), index)
        return c


def aligned_empty(n, dtype, align=64):
    """numpy array of n items whose data pointer is `align`-byte aligned."""
    itemsize = np.dtype(dtype).itemsize
    raw = np.empty(n * itemsize + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n * itemsize].view(dtype)


def aligned_copy(a, align=64):
    a = np.ascontiguousarray(a)
    o = aligned_empty(a.size, a.dtype, align)
    o[:] = a.ravel()
    return o


class Oracle(BlockCodec):
    """The scalar restatement plus its batched drivers."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build()
        super().__init__(ORACLE_SO, "orc_")
        L = self.lib
        L.orc_batch_pack.restype = None
        L.orc_batch_pack.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, u32, C.c_size_t]
        L.orc_batch_unpack.restype = None
        L.orc_batch_unpack.argtypes = [C.c_int, vp, vp, vp, vp, u32, vp, C.c_size_t]
        L.orc_batch_reduce.restype = None
        L.orc_batch_reduce.argtypes = [C.c_int, vp, vp, C.c_size_t, vp, vp, vp]
        L.orc_fnv1a64.restype = C.c_uint64
        L.orc_fnv1a64.argtypes = [vp, C.c_size_t]
        L.orc_compressedbytes.restype = C.c_int64
        L.orc_compressedbytes.argtypes = [C.c_int64, u32]
        L.cpu_driver_time.restype = C.c_double
        L.cpu_driver_time.argtypes = [C.c_int, vp, vp, vp, vp, u32, C.c_size_t, C.c_int, C.c_int]
        L.stream_compress_d1.restype = C.c_size_t
        L.stream_compress_d1.argtypes = [vp, vp, vp, C.c_size_t, u32, vp]
        L.stream_compress_plain.restype = C.c_size_t
        L.stream_compress_plain.argtypes = [vp, vp, vp, C.c_size_t, vp]
        L.stream_decode_d1.restype = None
        L.stream_decode_d1.argtypes = [vp, vp, C.c_size_t, u32, vp]
        L.stream_decode_plain.restype = None
        L.stream_decode_plain.argtypes = [vp, vp, C.c_size_t, vp]
        L.query_search_time.restype = C.c_double
        L.query_search_time.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp, vp, C.c_int]
        L.stream_walk.restype = C.c_size_t
        L.stream_walk.argtypes = [vp, C.c_size_t, C.c_size_t, vp, vp, vp]

    MODES = {"plain": 0, "d1": 1, "for": 2}

    def batch_pack(self, mode, x, bit=0, seeds=None, masked=False, bits=None, off=None, out_bytes=None):
        """x: uint32[nblocks*128]; returns uint8 payload (uniform: nblocks*16*bit bytes)."""
        nblocks = x.size // 128
        if out_bytes is None:
            out_bytes = nblocks * 16 * bit
        out = np.zeros(out_bytes, dtype=np.uint8)
        self.lib.orc_batch_pack(self.MODES[mode], int(masked), _ptr(seeds), _ptr(x), _ptr(out), _ptr(off),
                                _ptr(bits), bit, nblocks)
        return out

    def batch_unpack(self, mode, payload, nblocks, bit=0, seeds=None, bits=None, off=None):
        out = np.zeros(nblocks * 128, dtype=np.uint32)
        self.lib.orc_batch_unpack(self.MODES[mode], _ptr(seeds), _ptr(payload), _ptr(off), _ptr(bits), bit,
                                  _ptr(out), nblocks)
        return out

    def batch_maxbits(self, x):
        nblocks = x.size // 128
        o = np.zeros(nblocks, dtype=np.uint8)
        self.lib.orc_batch_reduce(0, None, _ptr(x), nblocks, _ptr(o), None, None)
        return o

    def batch_maxbitsd1(self, seeds, x):
        nblocks = x.size // 128
        o = np.zeros(nblocks, dtype=np.uint8)
        self.lib.orc_batch_reduce(1, _ptr(seeds), _ptr(x), nblocks, _ptr(o), None, None)
        return o

    def batch_minmax(self, x):
        nblocks = x.size // 128
        lo = np.zeros(nblocks, dtype=np.uint32)
        hi = np.zeros(nblocks, dtype=np.uint32)
        self.lib.orc_batch_reduce(2, None, _ptr(x), nblocks, None, _ptr(lo), _ptr(hi))
        return lo, hi

    def fnv1a64(self, a):
        a = np.ascontiguousarray(a)
        return int(self.lib.orc_fnv1a64(_ptr(a), a.nbytes))

    def compressedbytes64(self, n, bit):
        return int(self.lib.orc_compressedbytes(int(n), bit))

    # -- the caller loops of example.c; `codec` is the BlockCodec whose block routines run inside them -------
    def stream_compress(self, x, d1=True, initvalue=0, codec=None):
        """example.c:56-75 (d1) / :135-150 (plain) -> uint8 stream."""
        codec = codec or self
        nblocks = x.size // 128
        buf = np.zeros(nblocks * 513 + 16, dtype=np.uint8)
        if d1:
            n = self.lib.stream_compress_d1(C.cast(codec._maxbitsd1, vp), C.cast(codec._packwmd1, vp), _ptr(x), nblocks,
                                            int(initvalue), _ptr(buf))
        else:
            n = self.lib.stream_compress_plain(C.cast(codec._maxbits, vp), C.cast(codec._packwm, vp), _ptr(x), nblocks,
                                               _ptr(buf))
        return buf[:n].copy()

    def stream_decode(self, stream, nblocks, d1=True, initvalue=0, codec=None):
        """example.c:101-110 (d1) / :170-175 (plain)."""
        codec = codec or self
        out = np.zeros(nblocks * 128, dtype=np.uint32)
        stream = np.ascontiguousarray(stream)
        if d1:
            self.lib.stream_decode_d1(C.cast(codec._unpackd1, vp), _ptr(stream), nblocks, int(initvalue), _ptr(out))
        else:
            self.lib.stream_decode_plain(C.cast(codec._unpack, vp), _ptr(stream), nblocks, _ptr(out))
        return out

    def stream_walk(self, stream, max_blocks):
        """-> (records found, bits[max_blocks], arena offsets[max_blocks], status)."""
        stream = np.ascontiguousarray(stream)
        bits = np.zeros(max_blocks, dtype=np.uint8)
        off = np.zeros(max_blocks, dtype=np.uint64)
        status = C.c_int(0)
        n = self.lib.stream_walk(_ptr(stream), stream.size, max_blocks, _ptr(bits), _ptr(off), C.byref(status))
        return int(n), bits, off, status.value

    # -- 8- / 16-lane wire formats (one block of 32*lanes integers) ------------------------------------
    def lanes_maxbits(self, lanes, x):
        self.lib.orc_lanes_maxbits.restype = u32
        return int(self.lib.orc_lanes_maxbits(u32(lanes), _ptr(x)))

    def lanes_pack(self, lanes, x, bit, masked=False):
        o = np.zeros(4 * lanes * min(bit, 32), np.uint8)
        self.lib.orc_lanes_pack(u32(lanes), C.c_int(int(masked)), _ptr(x), _ptr(o), u32(bit))
        return o

    def lanes_unpack(self, lanes, comp, bit):
        c = np.concatenate([np.asarray(comp, np.uint8), np.zeros(64, np.uint8)])
        o = np.zeros(32 * lanes, np.uint32)
        self.lib.orc_lanes_unpack(u32(lanes), _ptr(c), _ptr(o), u32(bit))
        return o

    def time_search(self, seeds, payload, off, bits, query_blocks, keys, reps=3, codec=None):
        """One simdsearchwithlengthd1 call per query on one core -> (seconds, positions, values)."""
        codec = codec or self
        nq = keys.size
        pos = np.zeros(nq, np.int32)
        val = np.zeros(nq, np.uint32)
        t = self.lib.query_search_time(C.cast(codec._searchlend1, vp), _ptr(seeds), _ptr(payload), _ptr(off), _ptr(bits),
                                       _ptr(query_blocks), _ptr(keys), nq, _ptr(pos), _ptr(val), reps)
        return float(t), pos, val

    def time_blocks(self, kind, fn_ptr, seeds, payload, ints, bit, nblocks, nthreads, reps):
        """Best-of-reps wall seconds for one pass; kind 0 unpack seeded, 1 unpack plain,
        2 pack seeded, 3 pack plain.  fn_ptr is a ctypes function object."""
        addr = C.cast(fn_ptr, vp)
        return float(self.lib.cpu_driver_time(kind, addr, _ptr(seeds), _ptr(payload), _ptr(ints), bit, nblocks,
                                              nthreads, reps))


REF_AVX_SO = os.path.join(_HERE, "_ref", "libsimdcomp_ref_avx.so")


def have_ref_avx():
    """The reference's AVX2 / AVX-512 packers can run here: the file exists and the CPU has both ISAs."""
    if not os.path.exists(REF_AVX_SO):
        return False
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return " avx2" in flags and " avx512f" in flags


class ReferenceAvx:
    """avxpack / avx512pack families of the compiled reference (src/avxbitpacking.c, src/avx512bitpacking.c)."""

    def __init__(self):
        self.lib = C.CDLL(REF_AVX_SO)
        self.f = {}
        for lanes, pre in ((8, "avx"), (16, "avx512")):
            mb = getattr(self.lib, pre + "maxbits")
            mb.restype, mb.argtypes = u32, [vp]
            fs = [mb]
            for name in ("pack", "packwithoutmask", "unpack"):
                g = getattr(self.lib, pre + name)
                g.restype, g.argtypes = None, [vp, vp, u32]
                fs.append(g)
            self.f[lanes] = fs

    def lanes_maxbits(self, lanes, x):
        return int(self.f[lanes][0](_ptr(x)))

    def lanes_pack(self, lanes, x, bit, masked=False):
        o = aligned_empty(4 * lanes * bit + 64, np.uint8)
        o[:] = 0
        self.f[lanes][1 if masked else 2](_ptr(x), _ptr(o), bit)
        return o[:4 * lanes * bit].copy()

    def lanes_unpack(self, lanes, comp, bit):
        c = aligned_copy(np.concatenate([np.asarray(comp, np.uint8), np.zeros(64, np.uint8)]))
        o = np.zeros(32 * lanes, np.uint32)
        self.f[lanes][3](_ptr(c), _ptr(o), bit)
        return o


def have_ref():
    return os.path.exists(REF_SO)


def Reference():
    """The compiled reference (raises FileNotFoundError when it was not built)."""
    if not have_ref():
        raise FileNotFoundError(REF_SO)
    return BlockCodec(REF_SO, "")
