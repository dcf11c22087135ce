#!/usr/bin/env python3
"""Does compress() (widths, offsets scan, pack into the byte stream) get faster when it works through the input in
pieces that stay in the memory-side cache between the width pass and the pack pass?  The three calls per piece are
enqueued back to back with the piece's stream position known beforehand (from a pass over the whole input), so only
the effect on the kernels and the launch boundaries is measured."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simdcomp_amd import binding, codec  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n = 1 << 30
    nblk = n // 128
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    w = torch.minimum(torch.randint(0, 21, (nblk,), device=dev, generator=g), torch.randint(0, 21, (nblk,), device=dev, generator=g))
    gaps = torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device=dev, generator=g)
    gaps &= ((1 << w) - 1).to(torch.int32).repeat_interleave(128)
    ids = torch.cumsum(gaps, 0, dtype=torch.int32)
    del gaps, w
    seeds = torch.empty(nblk, dtype=torch.int32, device=dev)
    bits = codec.maxbitsd1(ids, initvalue=0, initvalues_out=seeds)
    off, total = codec.payload_offsets(bits)
    T = int(total.item())
    want = torch.empty(T + nblk + 16, dtype=torch.uint8, device=dev)
    codec.pack_stream(1, ids, bits, off, want, initvalue=0)
    h_off = off.cpu()
    L, P = binding.lib(), binding.ptr
    res = {"bits_per_int": round(T * 8 / n, 2)}
    out = torch.zeros_like(want)
    bits2 = torch.empty_like(bits)
    seeds2 = torch.empty_like(seeds)
    off2 = torch.empty_like(off)
    ws = torch.empty(max(8, L.simdcomp_hip_payload_offsets_workspace(nblk)), dtype=torch.uint8, device=dev)
    tot = torch.zeros(1, dtype=torch.int64, device=dev)

    def run(cb):
        st = torch.cuda.current_stream().cuda_stream
        for f in range(0, nblk, cb):
            c = min(cb, nblk - f)
            init = 0  # explicit seeds below make the chain's start irrelevant for the pack; widths need the true predecessor
            binding.check(L.simdcomp_hip_maxbitsd1(0, P(seeds) + 4 * f, ids.data_ptr() + 512 * f, c, bits2.data_ptr() + f, None, st), "w")
            binding.check(L.simdcomp_hip_payload_offsets(bits2.data_ptr() + f, c, off2.data_ptr() + 8 * f, P(tot), P(ws), st), "s")
            binding.check(L.simdcomp_hip_pack_stream(1, 0, 0, P(seeds) + 4 * f, ids.data_ptr() + 512 * f,
                                                     out.data_ptr() + int(h_off[f]) + f, off2.data_ptr() + 8 * f, bits2.data_ptr() + f, c, None, st), "p")

    for log2c in (23, 20, 19, 18, 17, 16, 15):
        cb = 1 << log2c
        out.zero_()
        run(cb)
        torch.cuda.synchronize()
        ok = bool(torch.equal(out[:T + nblk], want[:T + nblk]))
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(cb)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        res[f"piece_2^{log2c}_blocks_{(cb * 512) >> 20}MiB"] = {"ms": round(best, 4), "same_bytes": ok}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
