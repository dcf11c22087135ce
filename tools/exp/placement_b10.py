"""One process: the payload of one width moved inside an over-allocated buffer (and the output likewise): does the
3-5 point spread between processes follow the relative placement of payload and output?
    python tools/exp/placement_b10.py [widths=10,12,26]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simdcomp_amd import codec  # noqa: E402

PEAK = 8.0e12
n = 1 << 30
nblk = n // 128
dev = torch.device("cuda:0")
widths = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10,12,26").split(",")]
g = torch.Generator(device=dev)
g.manual_seed(3)
raw = torch.randint(0, 2 ** 31 - 1, (n,), dtype=torch.int32, device=dev, generator=g)
slack = 96 << 20
out_big = torch.empty(n + slack // 4, dtype=torch.int32, device=dev)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return min(ts)


res = {"out": hex(out_big.data_ptr())}
offsets = [0, 4096, 65536, 1 << 18] + [k << 20 for k in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 80)]
for b in widths:
    ids = torch.cumsum(raw & ((1 << b) - 1), 0, dtype=torch.int32)
    o = torch.empty(nblk, dtype=torch.int32, device=dev)
    big = torch.empty(nblk * 16 * b + slack, dtype=torch.uint8, device=dev)
    base = codec.packd1(ids, b, initvalue=0, masked=False, initvalues_out=o).clone()
    del ids
    rows = {}
    for po in offsets:
        big[po: po + nblk * 16 * b].copy_(base)
        t = timed(lambda: codec.unpackd1(o, big[po: po + nblk * 16 * b], nblk, b, out=out_big[:n]))
        rows[f"payload+{po >> 10}K"] = round(nblk * (516 + 16 * b) / t / PEAK, 4)
    big[: nblk * 16 * b].copy_(base)
    for oo in offsets[1:]:
        t = timed(lambda: codec.unpackd1(o, big[: nblk * 16 * b], nblk, b, out=out_big[oo // 4: oo // 4 + n]))
        rows[f"out+{oo >> 10}K"] = round(nblk * (516 + 16 * b) / t / PEAK, 4)
    res[str(b)] = {"payload": hex(big.data_ptr()), "min": min(rows.values()), "max": max(rows.values()), "rows": rows}
    del big, base, o
print(json.dumps(res))
