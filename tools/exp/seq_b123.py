#!/usr/bin/env python3
"""Decode rate of b = 1, 2, 3 (and 32 for the population) in the benchmark's order of launches (b = 1..32 per step)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simdcomp_amd import codec
PEAK = 8.0e12
dev = torch.device("cuda:0")
n = 1 << 30; nblk = n // 128
g = torch.Generator(device=dev); g.manual_seed(3)
raw = torch.randint(0, 2 ** 31 - 1, (n,), dtype=torch.int32, device=dev, generator=g)
out = torch.empty(n, dtype=torch.int32, device=dev)
pay, offs = {}, {}
widths = [int(x) for x in os.environ.get("SEQ_WIDTHS", "1,2,3,5,7,16,32").split(",")]
for b in widths:
    ids = torch.cumsum(raw & ((1 << b) - 1) if b < 31 else raw, 0, dtype=torch.int32)
    o = torch.empty(nblk, dtype=torch.int32, device=dev)
    pay[b] = codec.packd1(ids, b, initvalue=0, masked=False, initvalues_out=o); offs[b] = o
    del ids
acc = {b: [] for b in widths}
for step in range(8):
    evs = []
    for b in widths:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); codec.unpackd1(offs[b], pay[b], nblk, b, out=out); e1.record(); evs.append((b, e0, e1))
    torch.cuda.synchronize()
    if step >= 2:
        for b, e0, e1 in evs: acc[b].append(e0.elapsed_time(e1) * 1e-3)
print(os.environ.get("SIMDCOMP_HIP_LIB", "default").split("/")[-1], {b: round(nblk * (512 + 16 * b + 4) / (sum(v) / len(v)) / PEAK, 4) for b, v in acc.items()})
