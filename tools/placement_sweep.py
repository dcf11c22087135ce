#!/usr/bin/env python3
"""Why do the narrow widths of the headline decoder come out in two populations across boxes?  One process, one GPU:

  isolated   every width alone, 10 launches back to back (no other kernel in between)
  sequence   the benchmark's order b = 1..32 within a step, and the reverse order
  placement  the output buffer and the payload moved inside over-allocated buffers (byte offsets that are multiples
             of 4 KiB .. 2 MiB, and odd multiples of 256 B), for b in {1, 2, 3, 32}
  after      b = 1 timed right behind a launch of b = 32, of b = 16, of a fill, and behind an idle gap

Prints one JSON object; profiles/r02_placement.json keeps the runs."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simdcomp_amd import codec  # noqa: E402

PEAK = 8.0e12


def box_facts():
    """What sysfs says about the GPU this ran on (partition modes, clock tables, firmware), for telling boxes apart."""
    import glob

    facts = {"host": os.uname().nodename}
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        for name in ("current_compute_partition", "current_memory_partition", "pp_dpm_mclk", "pp_dpm_sclk", "pp_dpm_fclk",
                     "vbios_version", "unique_id", "power_dpm_force_performance_level", "mem_info_vram_total",
                     "mem_info_vram_used"):
            try:
                facts[os.path.basename(os.path.dirname(dev)) + "/" + name] = open(os.path.join(dev, name)).read().strip()
            except OSError:
                pass
    return facts


def main():
    log2n = int(os.environ.get("SWEEP_LOG2N", "30"))
    dev = torch.device("cuda:0")
    n = 1 << log2n
    nblk = n // 128
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    raw = torch.randint(0, 2 ** 31 - 1, (n,), dtype=torch.int32, device=dev, generator=g)
    slack = 4 << 20
    out_big = torch.empty(n + slack // 4, dtype=torch.int32, device=dev)
    widths = list(range(1, 33))
    pay, offs = {}, {}
    for b in widths:
        ids = torch.cumsum(raw & ((1 << b) - 1) if b < 31 else raw, 0, dtype=torch.int32)
        o = torch.empty(nblk, dtype=torch.int32, device=dev)
        big = torch.empty(nblk * 16 * b + slack, dtype=torch.uint8, device=dev)
        codec.packd1(ids, b, initvalue=0, masked=False, initvalues_out=o, out=big[: nblk * 16 * b])
        pay[b], offs[b] = big, o
        del ids
    del raw

    def frac(b, t):
        return round(nblk * (512 + 16 * b + 4) / t / PEAK, 4)

    def launch(b, out_off=0, pay_off=0):
        p = pay[b]
        if pay_off:
            p[pay_off: pay_off + nblk * 16 * b].copy_(p[: nblk * 16 * b]) if False else None
        codec.unpackd1(offs[b], p[pay_off: pay_off + nblk * 16 * b], nblk, b, out=out_big[out_off // 4: out_off // 4 + n])

    def timed(fn, reps=10):
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
        ts.sort()
        return ts[len(ts) // 2], ts[0]

    res = {"log2n": log2n, "addresses": {"out": hex(out_big.data_ptr()), **{f"payload_{b}": hex(pay[b].data_ptr()) for b in (1, 2, 3, 32)},
                                         **{f"offsets_{b}": hex(offs[b].data_ptr()) for b in (1, 2, 3, 32)}}}
    iso = {}
    for b in widths:
        med, best = timed(lambda: launch(b))
        iso[str(b)] = {"median": frac(b, med), "best": frac(b, best)}
    res["isolated"] = iso

    def sequence(order, steps=5):
        for b in order:
            launch(b)
        torch.cuda.synchronize()
        acc = {b: [] for b in order}
        for _ in range(steps):
            evs = []
            for b in order:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch(b)
                e1.record()
                evs.append((b, e0, e1))
            torch.cuda.synchronize()
            for b, e0, e1 in evs:
                acc[b].append(e0.elapsed_time(e1) * 1e-3)
        return {str(b): frac(b, sum(v) / len(v)) for b, v in acc.items()}

    res["sequence_1_to_32"] = sequence(widths)
    res["sequence_32_to_1"] = sequence(widths[::-1])
    res["sequence_1_to_32_again"] = sequence(widths)

    place = {}
    for b in (() if os.environ.get("SWEEP_QUICK") else (1, 2, 3, 32)):
        # the payload copy inside its buffer: move it once per offset (the copy is outside the timed launches)
        base = pay[b][: nblk * 16 * b].clone()
        rows = {}
        for pay_off in (0, 256, 4096, 65536, 1 << 20, (2 << 20) + 4096):
            pay[b][pay_off: pay_off + nblk * 16 * b].copy_(base)
            for out_off in (0, 256, 4096, 65536, 1 << 20, (2 << 20) + 4096):
                med, best = timed(lambda: launch(b, out_off, pay_off), reps=6)
                rows[f"payload+{pay_off},out+{out_off}"] = frac(b, med)
        pay[b][: nblk * 16 * b].copy_(base)
        del base
        vals = list(rows.values())
        place[str(b)] = {"min": min(vals), "max": max(vals), "rows": rows}
    res["placement"] = place

    after = {}
    fillbuf = out_big[:n]

    def pair(first):
        def f():
            first()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch(1)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) * 1e-3
        f()
        ts = sorted(f() for _ in range(8))
        return frac(1, ts[len(ts) // 2])

    after["b=32"] = pair(lambda: launch(32))
    after["b=16"] = pair(lambda: launch(16))
    after["b=1"] = pair(lambda: launch(1))
    after["fill"] = pair(lambda: fillbuf.fill_(1))
    after["idle 20 ms"] = pair(lambda: (torch.cuda.synchronize(), time.sleep(0.02)))
    res["b=1_after"] = after
    res["box"] = box_facts()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
