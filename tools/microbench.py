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


def mixed(n, nblk, ints, out, reps):
    """Mixed-width d1 stream (widths 1..20 skewed small, the example.c convention):
    width reduction, offset scan, pack, unpack."""
    dev = ints.device
    widths = torch.minimum(torch.randint(1, 21, (nblk,), device=dev), torch.randint(1, 21, (nblk,), device=dev))
    gaps = torch.randint(0, 2 ** 31 - 1, (n,), dtype=torch.int32, device=dev)
    gaps &= ((1 << widths.to(torch.int64)) - 1).to(torch.int32).repeat_interleave(128)
    ids = torch.cumsum(gaps, 0, dtype=torch.int32)
    del gaps
    seeds = torch.empty(nblk, dtype=torch.int32, device=dev)
    bits = codec.maxbitsd1(ids, initvalue=0, initvalues_out=seeds)
    off, total = codec.payload_offsets(bits)
    nbytes = int(total.item())
    payload = torch.empty(nbytes + 16, dtype=torch.uint8, device=dev)
    codec.pack_mixed(1, ids, bits, off, payload, initvalues=seeds)
    codec.unpack_mixed(1, payload, bits, off, nblk, initvalues=seeds, out=out)
    assert torch.equal(out, ids)
    extra = nblk * (4 + 1 + 8)
    rows = {
        "maxbitsd1": (lambda: codec.maxbitsd1(ids, initvalue=0, initvalues_out=seeds), nblk * (512 + 5)),
        "payload_offsets": (lambda: codec.payload_offsets(bits), nblk * 9),
        "pack_mixed": (lambda: codec.pack_mixed(1, ids, bits, off, payload, initvalues=seeds), nblk * 512 + nbytes + extra),
        "unpack_mixed": (lambda: codec.unpack_mixed(1, payload, bits, off, nblk, initvalues=seeds, out=out),
                         nblk * 512 + nbytes + extra),
    }
    res = {"avg_bits": round(nbytes / nblk / 16, 2)}
    for name, (f, by) in rows.items():
        t = timeit(f, reps)
        res[name] = {"ms": round(t * 1e3, 4), "Gint/s": round(n / t / 1e9, 1), "GB/s": round(by / t / 1e9, 1),
                     "frac": round(by / t / PEAK, 3)}
    print("mixed", json.dumps(res), flush=True)


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
        if what == "mixed":
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
    if "mixed" in a.what.split(","):
        mixed(n, nblk, ints, out, a.reps)


if __name__ == "__main__":
    main()
