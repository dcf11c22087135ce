#!/usr/bin/env python3
"""Host API on ordinary (pageable) caller memory: simdunpack_length staged (the runtime bounces pageable copies through
its own pinned buffers) against pinning the caller's buffers for the duration of the call (SIMDCOMP_HIP_REGISTER=1:
hipHostRegister, kernel in place, hipHostUnregister).  Run once with and once without the variable."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simdcomp_amd import binding  # noqa: E402


def main():
    L = binding.lib()
    res = {"SIMDCOMP_HIP_REGISTER": os.environ.get("SIMDCOMP_HIP_REGISTER")}
    for log2n in (22, 24, 26, 28):
        n = 1 << log2n
        for bit in (8, 32):
            nbytes = L.simdpack_compressedbytes(n, bit)
            x = (np.arange(n, dtype=np.uint32) * np.uint32(2654435761)) & np.uint32((1 << bit) - 1 if bit < 32 else 0xFFFFFFFF)
            buf = np.zeros(nbytes + 64, np.uint8)
            out = np.zeros(n, np.uint32)
            L.simdpack_length(x.ctypes.data, n, buf.ctypes.data, bit)
            best = 1e9
            for _ in range(4):
                t0 = time.perf_counter()
                L.simdunpack_length(buf.ctypes.data, n, out.ctypes.data, bit)
                best = min(best, time.perf_counter() - t0)
            assert np.array_equal(out, x)
            res[f"2^{log2n}_b{bit}"] = {"ms": round(best * 1e3, 2), "Gint/s": round(n / best / 1e9, 2)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
