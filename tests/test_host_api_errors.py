"""The reference-signature host routines have no error channel (include/simdbitpacking.h:24-68 return void or a
pointer).  When the GPU path cannot run they must fail loudly -- never compute on the CPU instead: by default a message
on stderr and abort(), or the handler installed with simdcomp_hip_set_error_handler.  Runs in child processes (the
failure ends the process); needs no GPU: an impossible SIMDCOMP_HIP_DEVICES entry, or the absence of any device,
triggers it."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import ctypes as C, os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    from simdcomp_amd import binding
    lib = binding.lib()
    HANDLER = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)
    def on_error(msg, arg):
        sys.stdout.write("handler: " + msg.decode() + "\\n")
        sys.stdout.flush()
        os._exit(42)
    h = HANDLER(on_error)
    if {install}:
        lib.simdcomp_hip_set_error_handler(C.cast(h, C.c_void_p), None)
    x = np.arange(128, dtype=np.uint32)
    out = np.zeros(512, np.uint8)
    lib.simdpack(x.ctypes.data, out.ctypes.data, 7)
    print("returned", out[:4].tolist())
""")


def run_child(install):
    env = dict(os.environ, SIMDCOMP_HIP_DEVICES="63")  # no box has 64 GPUs: the device list is rejected
    return subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, install=install)], capture_output=True, text=True,
                          timeout=300, env=env)


def test_failure_goes_to_the_installed_handler():
    out = run_child(True)
    assert out.returncode == 42, (out.returncode, out.stdout, out.stderr[-500:])
    assert "handler: simdcomp_hip:" in out.stdout and "returned" not in out.stdout


def test_failure_without_handler_aborts_loudly():
    out = run_child(False)
    assert out.returncode not in (0, 42) and "simdcomp_hip:" in out.stderr and "returned" not in out.stdout
