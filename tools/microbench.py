#!/usr/bin/env python3
"""Per-width kernel timings on one GPU (device-resident), next to plain
copy/fill ceilings measured the same way.  Development tool, not the graded
benchmark (that is bench.py).

    python tools/microbench.py [--log2n 28] [--what unpackd1,unpack,pack,...] [--reps 5]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from simdcomp_amd import codec  # noqa: E402

PEAK = 8.0e12


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=28)
    ap.add_argument("--what", default="ceil,unpackd1,unpack,unpackFOR,packd1,pack,packFOR,maxbitsd1")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--bits", default="1,2,3,4,5,6,7,8,9,10,12,14,16,17,20,24,28,31,32")
    a = ap.parse_args()
    n = 1 << a.log2n
    nblk = n // 128
    bits = [int(b) for b in a.bits.split(",")]
    dev = torch.device("cuda:0")
    ints = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device=dev)
    out = torch.empty_like(ints)
    seeds = torch.randint(-2**31, 2**31 - 1, (nblk,), dtype=torch.int32, device=dev)
    payload = torch.randint(0, 255, (nblk * 16 * 32,), dtype=torch.uint8, device=dev)
    res = {}
    for what in a.what.split(","):
        if what == "ceil":
            t = timeit(lambda: out.copy_(ints), a.reps)
            res["copy"] = {"GB/s": 8 * n / t / 1e9}
            t = timeit(lambda: out.fill_(7), a.reps)
            res["fill"] = {"GB/s": 4 * n / t / 1e9}
            t = timeit(lambda: ints.sum(), a.reps)
            res["read(sum)"] = {"GB/s": 4 * n / t / 1e9}
            print(json.dumps(res), flush=True)
            continue
        row = {}
        for b in bits:
            if what == "unpackd1":
                f = lambda: codec.unpackd1(seeds, payload, nblk, b, out=out)
            elif what == "unpack":
                f = lambda: codec.unpack(payload, nblk, b, out=out)
            elif what == "unpackFOR":
                f = lambda: codec.unpackFOR(seeds, payload, nblk, b, out=out)
            elif what == "packd1":
                f = lambda: codec.packd1(ints, b, initvalues=seeds, out=payload)
            elif what == "pack":
                f = lambda: codec.pack(ints, b, masked=True, out=payload)
            elif what == "packFOR":
                f = lambda: codec.packFOR(seeds, ints, b, out=payload)
            elif what == "maxbitsd1":
                f = lambda: codec.maxbitsd1(ints, initvalues=seeds)
            else:
                raise SystemExit(what)
            t = timeit(f, a.reps)
            nbytes = nblk * (512 + 16 * b + (4 if "d1" in what or "FOR" in what else 0))
            if what == "maxbitsd1":
                nbytes = nblk * (512 + 4 + 1)
            row[b] = {"ms": round(t * 1e3, 4), "Gint/s": round(n / t / 1e9, 1), "GB/s": round(nbytes / t / 1e9, 1),
                      "frac": round(nbytes / t / PEAK, 3)}
            if what == "maxbitsd1":
                break
        res[what] = row
        print(what, json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
