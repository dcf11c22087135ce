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


def run(name, seconds):
    path = os.path.join(REF_BIN, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} was not built (needs the reference tree at build time)")
    p = subprocess.run([path], cwd=REF_BIN, capture_output=True, text=True, timeout=seconds)
    return p.returncode, p.stdout + p.stderr


def cpu_has(*flags):
    try:
        text = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return all(f" {f}" in text for f in flags)


@pytest.mark.timeout(400)
def test_reference_unit_tests_pass_against_the_library():
    """tests/unit.c: pack/unpack round trips at every width and length, FOR, d1, set/select/search; with the AVX2 and
    AVX-512 block tests as well when this host can execute them."""
    wide = cpu_has("avx2", "avx512f")
    rc, out = run("unit_hip_avx" if wide else "unit_hip", 380)
    assert rc == 0, out[-2000:]
    assert "All tests OK!" in out, out[-2000:]
    assert "bug" not in out.lower()


@pytest.mark.timeout(200)
def test_reference_unit_chars_passes_against_the_library():
    """tests/unit_chars.c: byte-shifted (unaligned) compressed buffers."""
    rc, out = run("unit_chars_hip", 180)
    assert rc == 0 and "Code looks good." in out, out[-2000:]


@pytest.mark.timeout(200)
def test_reference_example_runs_against_the_library():
    """example.c: simple demo, varying bit width demo and the compress()/decode loop."""
    rc, out = run("example_hip", 180)
    assert rc == 0 and out.count("Code works!") >= 2 and "bug" not in out, out[-2000:]
