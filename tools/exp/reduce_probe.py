#!/usr/bin/env python3
"""Development probe: the reductions at 2^30 integers (per-block maxbits / maxbitsd1 / maxmin, whole-array reduce,
checksum), as fractions of the HBM peak."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simdcomp_amd import codec  # noqa: E402

n = 1 << 30
dev = torch.device("cuda:0")
x = torch.randint(0, 1 << 20, (n,), dtype=torch.int32, device=dev)
seeds = torch.zeros(n // 128, dtype=torch.int32, device=dev)


def t(f, reps=7):
    f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


rows = {
    "maxbits": lambda: codec.maxbits(x),
    "maxbitsd1_chained": lambda: codec.maxbitsd1(x, initvalue=0),
    "maxbitsd1_seeds": lambda: codec.maxbitsd1(x, initvalues=seeds),
    "maxmin": lambda: codec.maxmin(x),
    "reduce_length": lambda: codec.reduce_length(x),
    "checksum": lambda: codec.checksum(x),
}
res = {}
for name, f in rows.items():
    ms = t(f)
    res[name] = {"ms": round(ms, 4), "frac": round(4 * n / (ms * 1e-3) / 8e12, 3)}
print("reduce", json.dumps(res))
