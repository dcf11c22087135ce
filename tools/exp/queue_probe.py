#!/usr/bin/env python3
"""Does the slow / fast state of a width belong to the stream (hardware queue) the kernel is launched on?  One
process, one set of buffers, the decode of a few widths timed on the default stream and on several new streams.

    python tools/exp/queue_probe.py [--bits 10,1,24,12] [--streams 8]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simdcomp_amd import binding, codec  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bits", default="10,1,24,12")
    ap.add_argument("--streams", type=int, default=8)
    ap.add_argument("--log2n", type=int, default=30)
    a = ap.parse_args()
    n = 1 << a.log2n
    nblk = n // 128
    dev = torch.device("cuda:0")
    bits = [int(b) for b in a.bits.split(",")]
    out = torch.empty(n, dtype=torch.int32, device=dev)
    corp = {}
    g = torch.Generator(device=dev)
    for b in bits:
        g.manual_seed(b)
        gaps = torch.randint(0, 2 ** 31 - 1, (n,), dtype=torch.int32, device=dev, generator=g) & ((1 << b) - 1)
        ids = torch.cumsum(gaps, 0, dtype=torch.int32)
        del gaps
        offs = torch.empty(nblk, dtype=torch.int32, device=dev)
        corp[b] = (codec.packd1(ids, b, initvalue=0, masked=False, initvalues_out=offs), offs)
        del ids
    torch.cuda.synchronize()
    L, P = binding.lib(), binding.ptr
    streams = [torch.cuda.default_stream()] + [torch.cuda.Stream() for _ in range(a.streams)]
    # isolated launches (synchronised after each) against five launches back to back, default stream
    st = streams[0]
    for b in bits:
        pay, offs = corp[b]
        res = []
        for batch in (1, 5, 1, 5):
            best = 1e9
            for r in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(batch):
                    binding.check(L.simdcomp_hip_unpackd1(P(offs), P(pay), P(out), nblk, b, st.cuda_stream), "unpackd1")
                e1.record(st)
                e1.synchronize()
                if r:
                    best = min(best, e0.elapsed_time(e1) / batch)
            res.append(nblk * (516 + 16 * b) / (best * 1e-3) / 8e12 * 100)
        print("b = %d: isolated %.1f, x5 %.1f, isolated %.1f, x5 %.1f" % (b, *res), flush=True)
    # the sequence of the benchmark: all widths back to back, per-width events
    for rnd in range(3):
        evs = []
        for b in bits:
            pay, offs = corp[b]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            binding.check(L.simdcomp_hip_unpackd1(P(offs), P(pay), P(out), nblk, b, st.cuda_stream), "unpackd1")
            e1.record(st)
            evs.append((b, e0, e1))
        torch.cuda.synchronize()
        print("sequence: " + " ".join("%d:%.1f" % (b, nblk * (516 + 16 * b) / (e0.elapsed_time(e1) * 1e-3) / 8e12 * 100) for b, e0, e1 in evs), flush=True)
    print("rows = stream (0 = default), columns = widths %s, %% of HBM peak" % bits)
    for rnd in range(2):
        for si, st in enumerate(streams):
            row = []
            for b in bits:
                pay, offs = corp[b]
                best = 1e9
                for r in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    binding.check(L.simdcomp_hip_unpackd1(P(offs), P(pay), P(out), nblk, b, st.cuda_stream), "unpackd1")
                    e1.record(st)
                    e1.synchronize()
                    if r:
                        best = min(best, e0.elapsed_time(e1))
                row.append(nblk * (516 + 16 * b) / (best * 1e-3) / 8e12 * 100)
            print("round %d stream %d: " % (rnd, si) + " ".join("%5.1f" % x for x in row), flush=True)


if __name__ == "__main__":
    main()
