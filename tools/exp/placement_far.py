#!/usr/bin/env python3
"""b = 1 and b = 32 decode rate against the distance between the output buffer and the payload / offsets, in steps
of 8 MiB over 1 GiB (one process, buffers over-allocated)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simdcomp_amd import codec
PEAK = 8.0e12
dev = torch.device("cuda:0")
n = 1 << 30; nblk = n // 128
g = torch.Generator(device=dev); g.manual_seed(3)
raw = torch.randint(0, 2 ** 31 - 1, (n,), dtype=torch.int32, device=dev, generator=g)
GiB = 1 << 30
out_big = torch.empty((4 * GiB + GiB) // 4, dtype=torch.int32, device=dev)
res = {}
for b in (1, 32):
    ids = torch.cumsum(raw & ((1 << b) - 1) if b < 31 else raw, 0, dtype=torch.int32)
    o = torch.empty(nblk, dtype=torch.int32, device=dev)
    big = torch.empty(nblk * 16 * b + GiB, dtype=torch.uint8, device=dev)
    codec.packd1(ids, b, initvalue=0, masked=False, initvalues_out=o, out=big[: nblk * 16 * b])
    del ids
    base = big[: nblk * 16 * b].clone()
    def rate(out_off, pay_off):
        p = big[pay_off: pay_off + nblk * 16 * b]
        ov = out_big[out_off // 4: out_off // 4 + n]
        codec.unpackd1(o, p, nblk, b, out=ov); torch.cuda.synchronize()
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); codec.unpackd1(o, p, nblk, b, out=ov); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        ts.sort()
        return round(nblk * (512 + 16 * b + 4) / ts[len(ts) // 2] / PEAK, 4)
    r = {"out_ptr": hex(out_big.data_ptr()), "payload_ptr": hex(big.data_ptr()), "offsets_ptr": hex(o.data_ptr())}
    r["out_offset_MiB"] = {str(k * 8): rate(k * (8 << 20), 0) for k in range(0, 128, 1)}
    pr = {}
    for k in range(0, 128, 4):
        off = k * (8 << 20)
        big[off: off + nblk * 16 * b].copy_(base)
        pr[str(k * 8)] = rate(0, off)
    big[: nblk * 16 * b].copy_(base)
    r["payload_offset_MiB"] = pr
    res[str(b)] = r
    del big, base
print(json.dumps(res))
