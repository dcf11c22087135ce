"""The host-only part of the library under AddressSanitizer + UndefinedBehaviorSanitizer (the reference has the same
switch: `make DEBUG=1`, its Makefile:5-6).  `make -C simdcomp_amd/csrc asan` compiles scalar_access.cpp and
host_arith.cpp with the host compiler and both sanitizers; tests/host_sanitizer_driver.cpp (single-value accessors on
exactly-sized heap buffers at every width, size arithmetic up to 2^40, the partitioners' edge cases: more parts than
blocks, nothing to split, offsets near 2^62) and the scalar restatement it checks against are compiled the same way and
linked to it.  CPU only: no HIP, no GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O1", "-ggdb", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-msse4.1"]


def test_host_routines_under_asan_and_ubsan(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no host compiler")
    csrc = os.path.join(ROOT, "simdcomp_amd", "csrc")
    made = subprocess.run(["make", "-C", csrc, "asan"], capture_output=True, text=True)
    assert made.returncode == 0, made.stdout[-2000:] + made.stderr[-2000:]
    orc_o = tmp_path / "oracle.o"
    exe = tmp_path / "host_sanitizer_driver"
    cc = subprocess.run(["gcc", *FLAGS, "-c", os.path.join(ROOT, "oracle", "simdcomp_oracle.c"), "-o", str(orc_o)],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    # the sanitizer runtime is linked statically so that the program does not depend on the order in which the
    # dynamic loader brings libraries in
    ld = subprocess.run(["g++", *FLAGS, "-std=c++17", "-static-libasan", "-static-libubsan",
                         os.path.join(ROOT, "tests", "host_sanitizer_driver.cpp"), str(orc_o),
                         os.path.join(csrc, "build", "libsimdcomp_host_asan.a"), "-o", str(exe)], capture_output=True, text=True)
    if ld.returncode != 0 and ("libasan" in ld.stderr or "libubsan" in ld.stderr):
        pytest.skip("the sanitizer runtimes are not installed: " + ld.stderr[-300:])
    assert ld.returncode == 0, ld.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    text = run.stdout + run.stderr
    assert run.returncode == 0, text[-4000:]
    assert "checks passed" in run.stdout
    assert "AddressSanitizer" not in text and "runtime error" not in text, text[-4000:]
