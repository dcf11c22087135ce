#!/usr/bin/env python3
"""Which buffer carries the slow / fast state of a width?  One allocation; K copies of the payload (and offsets) of
width b and K output buffers at different places in it; the decode rate of every (payload copy, output buffer) pair.

    python tools/exp/pair_probe.py [--bits 10,1,24] [--k 6] [--log2n 30]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simdcomp_amd import codec  # noqa: E402

TWO_MIB = 2 << 20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bits", default="10,1,24")
    ap.add_argument("--k", type=int, default=6)
    ap.add_argument("--log2n", type=int, default=30)
    a = ap.parse_args()
    n = 1 << a.log2n
    nblk = n // 128
    dev = torch.device("cuda:0")
    bits = [int(b) for b in a.bits.split(",")]
    up = lambda x: (x + TWO_MIB - 1) // TWO_MIB * TWO_MIB
    K = a.k
    # layout: K outputs interleaved with K payload copies of each width, so that copies lie 5+ GiB apart
    per_round = up(4 * n) + sum(up(16 * b * nblk) + up(4 * nblk) for b in bits)
    arena = torch.empty(K * per_round + 2 * TWO_MIB, dtype=torch.uint8, device=dev)
    at = (-arena.data_ptr()) % TWO_MIB
    outs, pays, offs = [], {b: [] for b in bits}, {b: [] for b in bits}
    for _ in range(K):
        outs.append(arena[at:at + 4 * n].view(torch.int32))
        at += up(4 * n)
        for b in bits:
            pays[b].append(arena[at:at + 16 * b * nblk])
            at += up(16 * b * nblk)
            offs[b].append(arena[at:at + 4 * nblk].view(torch.int32))
            at += up(4 * nblk)
    g = torch.Generator(device=dev)
    for b in bits:
        g.manual_seed(b)
        gaps = torch.randint(0, 2 ** 31 - 1, (n,), dtype=torch.int32, device=dev, generator=g) & ((1 << b) - 1)
        ids = torch.cumsum(gaps, 0, dtype=torch.int32)
        del gaps
        codec.packd1(ids, b, initvalue=0, masked=False, initvalues_out=offs[b][0], out=pays[b][0])
        del ids
        for j in range(1, K):
            pays[b][j].copy_(pays[b][0])
            offs[b][j].copy_(offs[b][0])
    torch.cuda.synchronize()

    def rate(b, j, i, reps=5):
        best = 1e9
        for r in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            codec.unpackd1(offs[b][j], pays[b][j], nblk, b, out=outs[i])
            e1.record()
            e1.synchronize()
            if r:
                best = min(best, e0.elapsed_time(e1))
        return nblk * (516 + 16 * b) / (best * 1e-3) / 8e12 * 100

    print("arena at 0x%x, round = %.2f GiB" % (arena.data_ptr(), per_round / 2 ** 30))
    for b in bits:
        print("b = %d: rows = payload copy, columns = output buffer (%% of HBM peak)" % b)
        for j in range(K):
            print("  " + " ".join("%5.1f" % rate(b, j, i) for i in range(K)), flush=True)


if __name__ == "__main__":
    main()
