#!/usr/bin/env python3
"""Timing probe: chained offsets and the in-place unpack of the byte stream run segment by segment on two streams, so
that the block sums of segment s + 1 overlap the unpack of segment s.  Timing only: every segment's chain starts at 0
(the carry between segments is not passed on), so the decoded values are wrong from the second segment on.

    python tools/exp/overlap_probe.py [--log2n 30] [--segments 1,2,4,8,16]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from simdcomp_amd import binding, codec  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=30)
    ap.add_argument("--segments", default="1,2,4,8,16")
    ap.add_argument("--reps", type=int, default=7)
    a = ap.parse_args()
    n = 1 << a.log2n
    nblk = n // 128
    dev = torch.device("cuda:0")
    torch.manual_seed(20260)
    w = torch.minimum(torch.randint(0, 21, (nblk,), device=dev), torch.randint(0, 21, (nblk,), device=dev))
    gaps = torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device=dev)
    gaps &= ((1 << w) - 1).to(torch.int32).repeat_interleave(128)
    ids = torch.cumsum(gaps, 0, dtype=torch.int32)
    del gaps
    seeds = torch.empty(nblk, dtype=torch.int32, device=dev)
    bits = codec.maxbitsd1(ids, initvalue=0, initvalues_out=seeds)
    off, total = codec.payload_offsets(bits)
    T = int(total.item())
    strm = torch.empty(T + nblk + 64, dtype=torch.uint8, device=dev)
    codec.pack_stream(1, ids, bits, off, strm, initvalue=0)
    del ids
    out = torch.empty(n, dtype=torch.int32, device=dev)
    chained = torch.empty(nblk, dtype=torch.int32, device=dev)
    L, P = binding.lib(), binding.ptr
    ws = torch.empty(2 * L.simdcomp_hip_chain_workspace(nblk) + 4096, dtype=torch.uint8, device=dev)
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    res = {}
    for nseg in [int(x) for x in a.segments.split(",")]:
        cuts = [nblk * i // nseg for i in range(nseg + 1)]
        ws_each = (L.simdcomp_hip_chain_workspace(cuts[1] - cuts[0] + 1) + 255) // 256 * 256
        evs = [torch.cuda.Event() for _ in range(nseg)]

        def run():
            fork = torch.cuda.Event()
            fork.record(main_s)
            side.wait_event(fork)
            for s in range(nseg):
                f, c = cuts[s], cuts[s + 1] - cuts[s]
                binding.check(L.simdcomp_hip_chain_offsets(0, P(strm) + f, P(off) + 8 * f, P(bits) + f, c, 1, P(chained) + 4 * f,
                                                           P(ws) + (s % 2) * ws_each, side.cuda_stream), "chain")
                evs[s].record(side)
                main_s.wait_event(evs[s])
                binding.check(L.simdcomp_hip_unpack_stream(1, P(chained) + 4 * f, P(strm) + f, P(off) + 8 * f, P(bits) + f,
                                                           P(out) + 512 * f, c, main_s.cuda_stream), "unpack")

        run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_s)
            run()
            e1.record(main_s)
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        res[nseg] = round(best, 4)
    print("overlap", json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
