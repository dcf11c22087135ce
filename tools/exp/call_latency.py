#!/usr/bin/env python3
"""Cost of one host-signature block call (simdpackwithoutmaskd1 / simdunpackd1 on one 128-integer block in host
memory) and of a small simdpack_length, through ctypes."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simdcomp_amd import binding  # noqa: E402

L = binding.lib()
x = np.cumsum(np.random.default_rng(1).integers(0, 100, 128)).astype(np.uint32)
out = np.zeros(4 * 32, np.uint32)
back = np.zeros(128, np.uint32)
vp = C.c_void_p
L.simdpackwithoutmaskd1.argtypes = [C.c_uint32, vp, vp, C.c_uint32]
L.simdpackwithoutmaskd1.restype = None
L.simdunpackd1.argtypes = [C.c_uint32, vp, vp, C.c_uint32]
L.simdunpackd1.restype = None
for _ in range(200):
    L.simdpackwithoutmaskd1(0, x.ctypes.data, out.ctypes.data, 7)
    L.simdunpackd1(0, out.ctypes.data, back.ctypes.data, 7)
assert np.array_equal(back, x)
n = 5000
t0 = time.perf_counter()
for _ in range(n):
    L.simdpackwithoutmaskd1(0, x.ctypes.data, out.ctypes.data, 7)
t1 = time.perf_counter()
for _ in range(n):
    L.simdunpackd1(0, out.ctypes.data, back.ctypes.data, 7)
t2 = time.perf_counter()
print("block call: pack %.1f us, unpack %.1f us" % ((t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6))
