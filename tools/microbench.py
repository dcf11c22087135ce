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

from simdcomp_amd import binding, codec  # noqa: E402

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


def stream(n, nblk, reps):
    """The example.c byte stream on the device (widths 0..20 skewed small, 2 % of the blocks 32-bit):
    build, record index, unbuild, chained offsets (the CPU loop of example.c:101-110 is timed by bench.py)."""
    import numpy as np

    dev = torch.device("cuda:0")
    torch.manual_seed(20260)  # the same stream in every run, so that runs of different builds compare
    w = torch.minimum(torch.randint(0, 21, (nblk,), device=dev), torch.randint(0, 21, (nblk,), device=dev))
    gaps = torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device=dev)
    gaps &= ((1 << w) - 1).to(torch.int32).repeat_interleave(128)
    ids = torch.cumsum(gaps, 0, dtype=torch.int32)
    del gaps
    wild = torch.randint(0, nblk, (nblk // 50,), device=dev)
    ids.view(-1, 128)[wild] = torch.randint(-2 ** 31, 2 ** 31 - 1, (wild.numel(), 128), dtype=torch.int32, device=dev)
    seeds = torch.empty(nblk, dtype=torch.int32, device=dev)
    bits = codec.maxbitsd1(ids, initvalue=0, initvalues_out=seeds)
    off, total = codec.payload_offsets(bits)
    T = int(total.item())
    arena = torch.empty(T + 16, dtype=torch.uint8, device=dev)
    codec.pack_mixed(1, ids, bits, off, arena, initvalues=seeds)
    strm = codec.stream_build(arena, bits, off, T)
    ibits, ioff, res = codec.stream_index(strm, nblk)
    check = not os.environ.get("MB_NOCHECK")  # (timing kernels that were cut down on purpose)
    assert not check or (res.tolist() == [nblk, T, 0] and torch.equal(ibits, bits) and torch.equal(ioff, off))
    arena2 = codec.stream_unbuild(strm, bits, off, T)
    chained = codec.chain_offsets(strm, bits, off, stream_layout=True)
    assert torch.equal(chained, seeds) and torch.equal(arena2[:T], arena[:T])
    strm2 = torch.empty_like(strm)
    codec.pack_stream(1, ids, bits, off, strm2, initvalue=0)
    assert torch.equal(strm2, strm)
    out = torch.empty_like(ids)
    ws_i = torch.empty(binding.lib().simdcomp_hip_stream_index_workspace(T + nblk), dtype=torch.uint8, device=dev)
    ws_c = torch.empty(binding.lib().simdcomp_hip_chain_workspace(nblk), dtype=torch.uint8, device=dev)
    r3 = torch.empty(3, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    L, P = binding.lib(), binding.ptr
    rows = {
        "stream_build": (lambda: L.simdcomp_hip_stream_build(P(arena), P(off), P(bits), nblk, P(strm), st), 2 * T + 10 * nblk),
        "stream_index": (lambda: L.simdcomp_hip_stream_index(P(strm), T + nblk, nblk, P(ibits), P(ioff), P(r3), P(ws_i), st),
                         T + 10 * nblk),
        "stream_unbuild": (lambda: L.simdcomp_hip_stream_unbuild(P(strm), P(off), P(bits), nblk, P(arena2), st), 2 * T + 10 * nblk),
        "chain_offsets": (lambda: L.simdcomp_hip_chain_offsets(0, P(strm), P(off), P(bits), nblk, 1, P(chained), P(ws_c), st),
                          T + 14 * nblk),
        "chain_offsets_arena": (lambda: L.simdcomp_hip_chain_offsets(0, P(arena), P(off), P(bits), nblk, 0, P(chained), P(ws_c), st),
                                T + 13 * nblk),
        "unpack_stream": (lambda: codec.unpack_stream(1, strm, bits, off, nblk, initvalues=seeds, out=out), 512 * nblk + T + 14 * nblk),
        "unpack_mixed": (lambda: codec.unpack_mixed(1, arena, bits, off, nblk, initvalues=seeds, out=out), 512 * nblk + T + 13 * nblk),
        "pack_stream": (lambda: codec.pack_stream(1, ids, bits, off, strm2, initvalue=0), 512 * nblk + T + 10 * nblk),
        "maxbitsd1": (lambda: codec.maxbitsd1(ids, initvalue=0, initvalues_out=seeds), 517 * nblk),
        "payload_offsets": (lambda: codec.payload_offsets(bits), 9 * nblk),
    }
    # compress() as one call
    ws_k = torch.empty(max(16, L.simdcomp_hip_stream_compress_workspace(nblk)), dtype=torch.uint8, device=dev)
    nb_k = torch.zeros(1, dtype=torch.int64, device=dev)
    strm3 = torch.empty(513 * nblk if nblk < (1 << 21) else T + nblk + 1024, dtype=torch.uint8, device=dev)
    binding.check(L.simdcomp_hip_stream_compress(1, 0, P(ids), nblk, P(strm3), P(nb_k), P(ws_k), st), "stream_compress")
    assert not check or (int(nb_k.item()) == T + nblk and torch.equal(strm3[:T + nblk], strm)), "stream_compress differs"
    rows["stream_compress"] = (lambda: L.simdcomp_hip_stream_compress(1, 0, P(ids), nblk, P(strm3), P(nb_k), P(ws_k), st),
                               512 * nblk + T + nblk)
    # the decode loop as one call (record index + chained decode straight from the bytes)
    ws_d = torch.empty(max(16, L.simdcomp_hip_stream_decode_workspace(T + nblk, nblk)), dtype=torch.uint8, device=dev)
    for mode, name in ((1, "stream_decode"), (0, "stream_decode_plain")):
        binding.check(L.simdcomp_hip_stream_decode(mode, 0, P(strm), T + nblk, nblk, P(out), P(r3), P(ws_d), st), name)
        assert not check or r3.tolist() == [nblk, T, 0], r3.tolist()
        if mode == 1:
            assert not check or torch.equal(out, ids), "stream_decode differs from the input"
        else:
            assert not check or torch.equal(out, codec.unpack_stream(0, strm, bits, off, nblk)), "plain stream_decode differs"
        rows[name] = ((lambda m: lambda: L.simdcomp_hip_stream_decode(m, 0, P(strm), T + nblk, nblk, P(out), P(r3), P(ws_d), st))(mode),
                      512 * nblk + T + nblk)
    res = {"avg_bits": round(T / nblk / 16, 2), "stream_MiB": round((T + nblk) / 2 ** 20, 1)}
    for name, (f, by) in rows.items():
        t = timeit(f, reps)
        res[name] = {"ms": round(t * 1e3, 4), "Gint/s": round(n / t / 1e9, 1), "GB/s": round(by / t / 1e9, 1), "frac": round(by / t / PEAK, 3)}
    whole = sum(res[k]["ms"] for k in ("stream_index", "chain_offsets", "unpack_stream")) * 1e-3
    res["decode_from_stream_bytes"] = {"ms": round(whole * 1e3, 4), "Gint/s": round(n / whole / 1e9, 1),
                                       "what": "stream_index + chain_offsets + unpack_stream (example.c:101-110)"}
    whole = sum(res[k]["ms"] for k in ("maxbitsd1", "payload_offsets", "pack_stream")) * 1e-3
    res["compress_to_stream"] = {"ms": round(whole * 1e3, 4), "Gint/s": round(n / whole / 1e9, 1),
                                 "what": "maxbitsd1 + payload_offsets + pack_stream (example.c:56-75)"}
    print("stream", json.dumps(res), flush=True)


def query(n, nblk, reps):
    """Batched search / select over a mixed-width d1 corpus: random block per query and one query per block
    (the reference's per-query search on one core is timed by bench.py)."""
    import numpy as np

    dev = torch.device("cuda:0")
    w = torch.minimum(torch.randint(1, 21, (nblk,), device=dev), torch.randint(1, 21, (nblk,), device=dev))
    gaps = torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device=dev)
    gaps &= ((1 << w) - 1).to(torch.int32).repeat_interleave(128)
    ids = torch.cumsum(gaps, 0, dtype=torch.int32)
    del gaps
    seeds = torch.empty(nblk, dtype=torch.int32, device=dev)
    bits = codec.maxbitsd1(ids, initvalue=0, initvalues_out=seeds)
    off, total = codec.payload_offsets(bits)
    T = int(total.item())
    arena = torch.empty(T + 16, dtype=torch.uint8, device=dev)
    codec.pack_mixed(1, ids, bits, off, arena, initvalues=seeds)
    res = {"blocks": nblk, "avg_bits": round(T / nblk / 16, 2)}
    rnd = torch.randint(0, nblk, (1 << 24,), device=dev)
    # "sorted_blocks": the same random batch in block order, what an intersection of posting lists issues (two
    # queries per block on average here); "hot_blocks": sorted, sixteen queries per block
    hot = torch.sort(torch.randint(0, nblk // 16, (1 << 24,), device=dev) * 16)[0]
    for name, nq, qb in (("random_blocks", 1 << 24, rnd), ("sorted_blocks", 1 << 24, torch.sort(rnd)[0]),
                         ("hot_blocks", 1 << 24, hot), ("every_block", nblk, None)):
        blk = qb if qb is not None else torch.arange(nblk, device=dev)
        keys = ids.view(-1, 128)[blk, torch.randint(0, 128, (nq,), device=dev)] + 1
        slots = torch.randint(0, 128, (nq,), dtype=torch.int32, device=dev)
        pos, val = codec.search(1, arena, keys, bits=bits, offsets=off, initvalues=seeds, query_blocks=qb)
        t = timeit(lambda: codec.search(1, arena, keys, bits=bits, offsets=off, initvalues=seeds, query_blocks=qb), reps)
        by = float((16 * bits[blk].long()).sum().item()) + nq * (13 + 8 + 4 + 8)
        res["search_" + name] = {"queries": nq, "ms": round(t * 1e3, 4), "Gquery/s": round(nq / t / 1e9, 2), "GB/s": round(by / t / 1e9, 1)}
        t = timeit(lambda: codec.select(1, arena, slots, bits=bits, offsets=off, initvalues=seeds, query_blocks=qb), reps)
        res["select_" + name] = {"queries": nq, "ms": round(t * 1e3, 4), "Gquery/s": round(nq / t / 1e9, 2)}
        if name in ("random_blocks", "sorted_blocks"):
            # the same with a block directory (one 16-byte metadata entry per block instead of three arrays)
            directory = codec.block_directory(bits, off, initvalues=seeds)
            dpos, dval = codec.search_directory(1, arena, directory, keys, query_blocks=qb)
            assert torch.equal(dpos, pos) and torch.equal(dval, val)
            t = timeit(lambda: codec.search_directory(1, arena, directory, keys, query_blocks=qb), reps)
            res["search_directory_" + name] = {"queries": nq, "ms": round(t * 1e3, 4), "Gquery/s": round(nq / t / 1e9, 2)}
            t = timeit(lambda: codec.select_directory(1, arena, directory, slots, query_blocks=qb), reps)
            res["select_directory_" + name] = {"queries": nq, "ms": round(t * 1e3, 4), "Gquery/s": round(nq / t / 1e9, 2)}
            t = timeit(lambda: codec.block_directory(bits, off, initvalues=seeds), reps)
            res["block_directory"] = {"blocks": nblk, "ms": round(t * 1e3, 4)}
    print("query", json.dumps(res), flush=True)


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
        if what in ("mixed", "stream", "query"):
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
            elif what in ("unpack8", "unpack16"):
                L = int(what[6:])
                f = lambda: codec.unpack_lanes(L, payload, n // (32 * L), b, out=out)
            elif what in ("pack8", "pack16"):
                L = int(what[4:])
                f = lambda: codec.pack_lanes(L, ints, b, out=payload)
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
    if "query" in a.what.split(","):
        query(n, nblk, a.reps)
    if "stream" in a.what.split(","):
        del ints, out, payload
        torch.cuda.empty_cache()
        stream(n, nblk, a.reps)


if __name__ == "__main__":
    main()
