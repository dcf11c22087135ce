#!/usr/bin/env python3
"""The launches whose hardware counters are kept under profiles/: run it under
`rocprofv3 --pmc ... --output-format csv -d DIR -- python3 tools/pmc_workload.py`, one
counter group per run, then `python tools/pmc_summary.py OUT.json DIR...`.

Every kernel is launched twice on resident data (2^30 integers): the encoders
simdpackwithoutmaskd1 at b = 1..32, the per-block-width encoder and decoder on a
mixed-width d1 corpus (widths 1..20), and the byte-stream kernels (record index,
chained offsets, in-place unpack).  Nothing is timed here."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simdcomp_amd import codec  # noqa: E402

LOG2N = int(os.environ.get("PMC_LOG2N", "30"))
WHAT = os.environ.get("PMC_WHAT", "encode,mixed,stream").split(",")


def main():
    dev = torch.device("cuda:0")
    n = 1 << LOG2N
    nblk = n // 128
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    raw = torch.randint(0, 2 ** 31 - 1, (n,), dtype=torch.int32, device=dev, generator=g)
    if "encode" in WHAT:
        offsets = torch.empty(nblk, dtype=torch.int32, device=dev)
        out = torch.empty(nblk * 16 * 32, dtype=torch.uint8, device=dev)
        for b in range(1, 33):
            gaps = raw & ((1 << b) - 1) if b < 31 else raw
            ids = torch.cumsum(gaps, 0, dtype=torch.int32)
            codec.packd1(ids, b, initvalue=0, masked=False, initvalues_out=offsets, out=out[: nblk * 16 * b])
            for _ in range(2):
                codec.packd1(ids, b, initvalues=offsets, masked=False, out=out[: nblk * 16 * b])
            del ids, gaps
        del out, offsets
    w = torch.minimum(torch.randint(1, 21, (nblk,), device=dev, generator=g), torch.randint(1, 21, (nblk,), device=dev, generator=g))
    ids = torch.cumsum(raw & ((1 << w) - 1).to(torch.int32).repeat_interleave(128), 0, dtype=torch.int32)
    del raw
    seeds = torch.empty(nblk, dtype=torch.int32, device=dev)
    bits = codec.maxbitsd1(ids, initvalue=0, initvalues_out=seeds)
    off, total = codec.payload_offsets(bits)
    T = int(total.item())
    arena = torch.empty(T + 16, dtype=torch.uint8, device=dev)
    back = torch.empty_like(ids)
    if "mixed" in WHAT or any(k.startswith("query") for k in WHAT):
        codec.pack_mixed(1, ids, bits, off, arena, initvalues=seeds, masked=False)
    if "mixed" in WHAT:
        for _ in range(2):
            codec.pack_mixed(1, ids, bits, off, arena, initvalues=seeds, masked=False)
            codec.unpack_mixed(1, arena, bits, off, nblk, initvalues=seeds, out=back)
    if "stream" in WHAT:
        strm = torch.empty(T + nblk, dtype=torch.uint8, device=dev)
        for _ in range(2):
            codec.pack_stream(1, ids, bits, off, strm, initvalue=0)
            codec.stream_index(strm, nblk)
            chained = codec.chain_offsets(strm, bits, off, stream_layout=True)
            codec.unpack_stream(1, strm, bits, off, nblk, initvalues=chained, out=back)
        assert torch.equal(back, ids)
    for kind in ("query_random", "query_sorted"):
        # 2^24 searches and selects, a random block each / the same batch in block order
        if kind in WHAT:
            qb = torch.randint(0, nblk, (1 << 24,), device=dev, generator=g)
            if kind == "query_sorted":
                qb = torch.sort(qb)[0]
            keys = ids.view(-1, 128)[qb, torch.randint(0, 128, (1 << 24,), device=dev, generator=g)] + 1
            slots = torch.randint(0, 128, (1 << 24,), dtype=torch.int32, device=dev, generator=g)
            for _ in range(2):
                codec.search(1, arena, keys, bits=bits, offsets=off, initvalues=seeds, query_blocks=qb)
                codec.select(1, arena, slots, bits=bits, offsets=off, initvalues=seeds, query_blocks=qb)
    torch.cuda.synchronize()
    print("pmc workload done", LOG2N, WHAT)


if __name__ == "__main__":
    main()
