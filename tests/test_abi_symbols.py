"""CPU-only checks of the C-ABI boundary: the shared library loads without a
GPU, exports every function the two headers declare, and its host-side
arithmetic (sizes, partitioner, single-value accessors) matches the oracle.
No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from simdcomp_amd import binding, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    text = re.sub(r"\btypedef\b[^;]*;", "", text)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_library_exports_every_declared_symbol():
    lib = binding.load()
    dev = declared_functions("simdcomp_hip.h")
    host = declared_functions("simdcomp.h")
    assert len(dev) >= 24 and len(host) >= 29
    for name in dev + host:
        assert hasattr(lib, name), f"{name} is declared in include/ but not exported"
    # and the binding tables cover the headers exactly
    assert sorted(binding.DEVICE_API) == dev
    assert sorted(binding.HOST_API) == host


def test_library_exports_nothing_else():
    """The dynamic symbol table holds the declared functions and nothing more: no helper of the library can collide
    with, or be interposed by, a symbol of the program that loads it (the reference's export list is just as closed,
    simdcomp.def.tpl:1-40)."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", binding.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1].split("@")[0] for line in out.splitlines() if line.strip())
    declared = sorted(set(declared_functions("simdcomp_hip.h") + declared_functions("simdcomp.h")))
    assert exported == declared, (sorted(set(exported) - set(declared)), sorted(set(declared) - set(exported)))


def test_soname_and_binary_substitution_resolve(tmp_path):
    """The library carries a soname, and a program that was compiled against the REFERENCE's headers and linked to the
    REFERENCE's own shared library (oracle/_ref/example_reflinked: NEEDED libsimdcomp.so, nothing of this repository
    in it) resolves every symbol against this library when libsimdcomp.so is a link to it: the dynamic loader's own
    trace (no code of the program runs, so no GPU is needed).  With the reference's library in its place the same
    binary runs here on the CPU and passes its own checks."""
    import subprocess

    dyn = subprocess.run(["readelf", "-d", binding.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "libsimdcomp_hip.so.0" in dyn and "SONAME" in dyn
    exe = os.path.join(ROOT, "oracle", "_ref", "example_reflinked")
    reflib = os.path.join(ROOT, "oracle", "_ref", "reflib")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/example_reflinked was not built (needs the reference tree at build time)")
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True, check=True).stdout
    assert "libsimdcomp.so" in needed and "simdcomp_hip" not in needed
    os.symlink(os.path.abspath(binding.LIB_PATH), tmp_path / "libsimdcomp.so")
    env = dict(os.environ, LD_LIBRARY_PATH=f"{tmp_path}:/opt/rocm/lib", LD_TRACE_LOADED_OBJECTS="1", LD_BIND_NOW="1", LD_WARN="1")
    out = subprocess.run([exe], capture_output=True, text=True, env=env)
    text = out.stdout + out.stderr
    assert f"libsimdcomp.so => {tmp_path}/libsimdcomp.so" in text, text
    assert "libamdhip64" in text and "undefined symbol" not in text and "not found" not in text, text
    # the control: the same binary on the library it was linked to
    env = dict(os.environ, LD_LIBRARY_PATH=reflib)
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0 and out.stdout.count("Code works!") >= 2, out.stdout[-1000:] + out.stderr[-1000:]


def test_version_and_sizes(orc):
    lib = binding.lib()
    assert lib.simdcomp_hip_version() >= 100
    for bit in range(33):
        for n in (0, 1, 3, 4, 5, 127, 128, 129, 130, 9999, 2 ** 30, 2 ** 33 + 5):
            assert lib.simdcomp_hip_compressedbytes(n, bit) == orc.compressedbytes64(n, bit)
        for n in (0, 1, 5, 130, 9999):
            assert lib.simdpack_compressedbytes(n, bit) == orc.compressedbytes(n, bit)
            assert lib.simdpackFOR_compressedbytes(n, bit) == orc.compressedbytesFOR(n, bit)
    assert lib.simdcomp_hip_compressedbytes(9999, 14) == 17504  # example.c prints this size
    for v in (0, 1, 2, 3, 255, 256, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF):
        assert lib.bits(v) == orc.bits(v)


def test_partition_block_ranges():
    lib = binding.lib()
    first, count = C.c_size_t(), C.c_size_t()
    for nblocks in (0, 1, 7, 8, 1000003, 2 ** 23, 125000000):
        for nparts in (1, 2, 3, 4, 8):
            end = 0
            sizes = []
            for p in range(nparts):
                assert lib.simdcomp_hip_partition(nblocks, nparts, p, C.byref(first), C.byref(count)) == 0
                assert first.value == end
                end = first.value + count.value
                sizes.append(count.value)
            assert end == nblocks and max(sizes) - min(sizes) <= 1
    assert lib.simdcomp_hip_partition(10, 0, 0, C.byref(first), C.byref(count)) < 0
    assert lib.simdcomp_hip_partition(10, 2, 2, C.byref(first), C.byref(count)) < 0


def test_partition_by_bytes_balances_cost():
    lib = binding.lib()
    nblk = 100000
    widths = synth.skewed_widths(5, nblk, 0, 32).astype(np.uint64)
    widths[: nblk // 2] = 1  # first half cheap, second half expensive
    off = np.concatenate([[0], np.cumsum(widths * 16)]).astype(np.uint64)
    first, count = C.c_size_t(), C.c_size_t()
    cost = lambda a, b: int(off[b] - off[a]) + 512 * (b - a)
    for nparts in (1, 2, 4, 8):
        end, costs = 0, []
        for p in range(nparts):
            assert lib.simdcomp_hip_partition_by_bytes(off.ctypes.data, nblk, nparts, p, C.byref(first),
                                                       C.byref(count)) == 0
            assert first.value == end
            end = first.value + count.value
            costs.append(cost(first.value, end))
        assert end == nblk
        assert max(costs) - min(costs) <= 2 * (512 + 512)  # within a couple of blocks of perfect balance


@pytest.mark.parametrize("bit", range(33))
def test_single_value_accessors(orc, bit):
    """simdselectFOR / simdsearchwithlengthFOR / simdfastset(FOR) are host routines of the library."""
    lib = binding.lib()
    x = np.sort(synth.random_bits(11, 128, bit))
    base = 1000
    xf = (x + np.uint32(base)).astype(np.uint32)
    comp = np.concatenate([orc.simdpackFOR(base, xf, bit), np.zeros(32, np.uint8)])
    for slot in range(128):
        assert lib.simdselectFOR(base, comp.ctypes.data, bit, slot) == int(xf[slot]) == \
            orc.simdselectFOR(base, comp, bit, slot)
    res = C.c_uint32()
    for key in list(xf[::9]) + [0, base, int(xf[-1]) + 1, 0xFFFFFFFF]:
        for length in (128, 77, 1):
            pos = lib.simdsearchwithlengthFOR(base, comp.ctypes.data, bit, length, int(key), C.byref(res))
            wpos, wval = orc.simdsearchwithlengthFOR(base, comp, bit, length, int(key))
            if wpos == 128 and bit:
                wval = 0  # the reference reads past the block there (the restatement reads the padding); the library does not
            assert (pos, res.value) == (wpos, wval)
    newv = synth.random_bits(12, 128, bit)
    mine, want = comp.copy(), comp[: 16 * bit].copy()
    for idx in (0, 1, 5, 31, 64, 126, 127):
        lib.simdfastset(mine.ctypes.data, bit, int(newv[idx]), idx)
        want = orc.simdfastset(want, bit, int(newv[idx]), idx)
        assert np.array_equal(mine[: 16 * bit], want)
        lib.simdfastsetFOR(base, mine.ctypes.data, bit, int(newv[idx]) + base, idx)
        want = orc.simdfastsetFOR(base, want, bit, int(newv[idx]) + base, idx)
        assert np.array_equal(mine[: 16 * bit], want)


def test_d1_single_block_accessors_golden(orc):
    """simdselectd1 / simdsearchwithlengthd1 / simdsearchd1 / simdscand1 of the library (host routines) against the
    answers the reference gave."""
    import golden_util as G
    import oracle

    lib = binding.lib()
    g = G.load("queries.npz")
    res = C.c_uint32()
    for bit in range(33):
        for kind in range(2):
            x, init = G.query_block(bit, kind)
            comp = oracle.aligned_copy(np.concatenate([orc.simdpackwithoutmaskd1(init, x, bit), np.zeros(528, np.uint8)])[:528])
            keys = G.query_keys(orc.simdunpackd1(init, comp[:16 * bit], bit), bit, kind)
            for i, slot in enumerate(G.QUERY_SLOTS):
                assert lib.simdselectd1(init, comp.ctypes.data, bit, slot) == g["select"][bit, kind, 0, i]
            for ki, key in enumerate(keys):
                for li, length in enumerate(G.QUERY_LENGTHS):
                    pos = lib.simdsearchwithlengthd1(init, comp.ctypes.data, bit, length, int(key), C.byref(res))
                    assert [pos, res.value] == g["search"][bit, kind, 0, ki, li].tolist()
                o4 = oracle.aligned_copy(np.full(4, init, np.uint32))
                pos = lib.simdsearchd1(o4.ctypes.data, comp.ctypes.data, bit, int(key), C.byref(res))
                assert [pos, res.value] + [int(v) for v in o4] == g["searchd1"][bit, kind, ki].tolist()
            o4 = oracle.aligned_copy(np.full(4, init, np.uint32))
            lib.simdscand1(o4.ctypes.data, comp.ctypes.data, bit)
            assert np.array_equal(o4, g["scan"][bit, kind])


@pytest.mark.parametrize("bit", [1, 4, 9, 17, 31, 32])
def test_fastsetd1(orc, ref, bit):
    """simdfastsetd1 / simdfastsetd1fromprevious rewrite one field exactly as the reference does."""
    import oracle

    lib = binding.lib()
    x = synth.sorted_gaps(5 + bit, 128, min(bit, 20), first=100)
    init = 100
    base = np.concatenate([orc.simdpackwithoutmaskd1(init, x, bit), np.zeros(528, np.uint8)])[:528]
    for idx in (0, 1, 63, 127):
        newv = int(x[idx]) + 1
        mine, theirs = oracle.aligned_copy(base), oracle.aligned_copy(base)
        lib.simdfastsetd1(init, mine.ctypes.data, bit, newv, idx)
        ref.lib.simdfastsetd1(C.c_uint32(init), C.c_void_p(theirs.ctypes.data), C.c_uint32(bit), C.c_uint32(newv), C.c_size_t(idx))
        assert np.array_equal(mine, theirs)
        prev = int(x[idx - 1]) if idx else init
        mine = oracle.aligned_copy(base)
        lib.simdfastsetd1fromprevious(mine.ctypes.data, bit, prev, newv, idx)
        assert np.array_equal(mine, theirs)


def test_public_headers_compile_as_plain_c_and_cxx(tmp_path):
    """include/simdcomp.h and include/simdcomp_hip.h are what a host language binds: they have to go through a C89
    compiler in pedantic mode (the reference is C89, Makefile:8) and through a C++ compiler, with no HIP header."""
    import shutil
    import subprocess

    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    src = '#include "simdcomp.h"\n#include "simdcomp_hip.h"\nint main(void) { return 0; }\n'
    c = tmp_path / "h.c"
    c.write_text(src)
    cc = shutil.which("gcc") or shutil.which("cc")
    assert cc, "no C compiler"
    for std in ("c89", "c99", "c11"):
        r = subprocess.run([cc, "-std=" + std, "-pedantic", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", str(c)],
                           capture_output=True, text=True)
        assert r.returncode == 0, (std, r.stderr)
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx:
        cpp = tmp_path / "h.cpp"
        cpp.write_text(src)
        r = subprocess.run([cxx, "-std=c++11", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", str(cpp)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_example_program_compiles_and_links(tmp_path):
    """examples/device_postings.c is the plain-C caller of the device API: it has to compile without warnings and link
    against the in-tree library with nothing but HIP's runtime header and library (no GPU needed for that)."""
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no HIP runtime header")
    cc = shutil.which("gcc") or shutil.which("cc")
    for std in ("c99", "c11", "gnu89"):
        r = subprocess.run([cc, "-std=" + std, "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                            "-I", os.path.join(root, "include"), os.path.join(root, "examples", "device_postings.c"),
                            "-L", os.path.join(root, "simdcomp_amd"), "-lsimdcomp_hip", "-L/opt/rocm/lib", "-lamdhip64",
                            "-Wl,-rpath," + os.path.join(root, "simdcomp_amd"), "-Wl,-rpath,/opt/rocm/lib",
                            "-o", str(tmp_path / ("example_" + std))], capture_output=True, text=True)
        assert r.returncode == 0, (std, r.stderr[-2000:])


def test_workspace_sizes_are_consistent():
    """The workspace-size functions are plain host arithmetic: multiples of 256, never shrinking with the input, large
    enough for what the kernels keep there (per 16 KiB chunk of stream: a table of 513 words and 1.5 logs of 1 KiB),
    and the one-call decode's workspace holds the index's and the chain's."""
    L = binding.lib()
    prev = 0
    for nbytes in (0, 1, 513, 16384, 16385, 1 << 20, (1 << 30) + 7, 1 << 36, 1 << 40):
        w = L.simdcomp_hip_stream_index_workspace(nbytes)
        chunks = (nbytes + 16383) // 16384
        assert w % 256 == 0 and w >= prev
        assert w >= chunks * (513 * 4 + 1536)
        assert w <= max(8192, nbytes) * 0.3 + (1 << 20), nbytes  # about a quarter of the stream
        prev = w
        nblk = nbytes // 100 + 1
        d = L.simdcomp_hip_stream_decode_workspace(nbytes, nblk)
        assert d >= w + L.simdcomp_hip_chain_workspace(nblk) + 13 * nblk
    prev = 0
    for nblk in (0, 1, 4095, 4096, 4097, 1 << 23, 1 << 29):
        for f in (L.simdcomp_hip_payload_offsets_workspace, L.simdcomp_hip_chain_workspace, L.simdcomp_hip_stream_compress_workspace):
            assert f(nblk) >= 0
        c = L.simdcomp_hip_stream_compress_workspace(nblk)
        assert c >= prev and c >= 9 * nblk
        prev = c
