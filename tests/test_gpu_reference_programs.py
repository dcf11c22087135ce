"""The reference's OWN programs judging the drop-in: its unit tests
(tests/unit.c, tests/unit_chars.c) and its example (example.c), compiled from
where they lie in the reference tree against include/simdcomp.h and linked to
libsimdcomp_hip.so (recipe: oracle/Makefile, target ref_programs; the binaries
travel under oracle/_ref/).  Every simdpack / simdunpack / *_length / FOR / d1
call they make is a round trip through the HIP kernels; select / search / set
are the library's host routines.  They print "All tests OK!", "Code looks
good." and "Code works!" only if every check of theirs passed."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
REF_BIN = os.path.join(HERE, "..", "oracle", "_ref")


LIB = os.path.abspath(os.path.join(HERE, "..", "simdcomp_amd", "libsimdcomp_hip.so"))


@pytest.fixture(scope="module")
def libdir(tmp_path_factory):
    """A directory for LD_LIBRARY_PATH that holds this library under its soname and under the two names of the
    reference's library (its Makefile:10-12,17-19: libsimdcomp.so.0.0.3 and the libsimdcomp.so link)."""
    d = tmp_path_factory.mktemp("libs")
    for name in ("libsimdcomp_hip.so.0", "libsimdcomp.so", "libsimdcomp.so.0.0.3"):
        os.symlink(LIB, d / name)
    return str(d)


def run(name, seconds, libdir):
    path = os.path.join(REF_BIN, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} was not built (needs the reference tree at build time)")
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([path], cwd=REF_BIN, capture_output=True, text=True, timeout=seconds, env=env)
    return p.returncode, p.stdout + p.stderr


def cpu_has(*flags):
    try:
        text = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return all(f" {f}" in text for f in flags)


@pytest.mark.timeout(400)
def test_reference_unit_tests_pass_against_the_library(libdir):
    """tests/unit.c: pack/unpack round trips at every width and length, FOR, d1, set/select/search; with the AVX2 and
    AVX-512 block tests as well when this host can execute them."""
    wide = cpu_has("avx2", "avx512f")
    rc, out = run("unit_hip_avx" if wide else "unit_hip", 380, libdir)
    assert rc == 0, out[-2000:]
    assert "All tests OK!" in out, out[-2000:]
    assert "bug" not in out.lower()


@pytest.mark.timeout(200)
def test_reference_unit_chars_passes_against_the_library(libdir):
    """tests/unit_chars.c: byte-shifted (unaligned) compressed buffers."""
    rc, out = run("unit_chars_hip", 180, libdir)
    assert rc == 0 and "Code looks good." in out, out[-2000:]


@pytest.mark.timeout(200)
def test_reference_example_runs_against_the_library(libdir):
    """example.c: simple demo, varying bit width demo and the compress()/decode loop."""
    rc, out = run("example_hip", 180, libdir)
    assert rc == 0 and out.count("Code works!") >= 2 and "bug" not in out, out[-2000:]


@pytest.mark.timeout(300)
def test_binaries_linked_to_the_reference_library_run_on_this_one(libdir):
    """Substitution without recompiling: example.c and tests/unit_chars.c compiled against the reference's OWN headers
    and linked to the reference's OWN libsimdcomp.so (oracle/Makefile, target ref_dropin; nothing of this repository
    is in those binaries) are started with libsimdcomp.so resolving to libsimdcomp_hip.so, and pass their checks on
    the GPU.  (The same binaries on the reference's library: tests/test_abi_symbols.py.)"""
    rc, out = run("example_reflinked", 180, libdir)
    assert rc == 0 and out.count("Code works!") >= 2 and "bug" not in out, out[-2000:]
    rc, out = run("unit_chars_reflinked", 180, libdir)
    assert rc == 0 and "Code looks good." in out, out[-2000:]
    # the process really ran on this library: the loader's trace of the same start
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib", LD_TRACE_LOADED_OBJECTS="1")
    trace = subprocess.run([os.path.join(REF_BIN, "example_reflinked")], capture_output=True, text=True, env=env).stdout
    assert f"libsimdcomp.so => {libdir}/libsimdcomp.so" in trace and "libamdhip64" in trace, trace


@pytest.mark.timeout(200)
@pytest.mark.parametrize("nblocks", ["8192", "100003"])
def test_device_api_example_program(libdir, nblocks):
    """examples/device_postings.c (plain C against include/simdcomp_hip.h): compress, decode and search a posting
    list with one call per step; the program checks its results against the input itself."""
    path = os.path.join(HERE, "..", "examples", "build", "device_postings")
    if not os.path.exists(path):
        pytest.skip("examples/build/device_postings is not built (make -C examples; build() of __graft_entry__ does it)")
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([path, nblocks], capture_output=True, text=True, timeout=180, env=env)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (p.stdout + p.stderr)[-2000:]
