#!/usr/bin/env python3
"""The one-call stream decode / compress captured in a HIP graph and replayed, against plain calls."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simdcomp_amd import binding, codec  # noqa: E402

n = 1 << 30
nblk = n // 128
dev = torch.device("cuda:0")
torch.manual_seed(20260)
w = torch.minimum(torch.randint(0, 21, (nblk,), device=dev), torch.randint(0, 21, (nblk,), device=dev))
gaps = torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device=dev)
gaps &= ((1 << w) - 1).to(torch.int32).repeat_interleave(128)
ids = torch.cumsum(gaps, 0, dtype=torch.int32)
del gaps
L, P = binding.lib(), binding.ptr
ws_k = torch.empty(L.simdcomp_hip_stream_compress_workspace(nblk), dtype=torch.uint8, device=dev)
nb = torch.zeros(1, dtype=torch.int64, device=dev)
strm = torch.empty(n // 128 * 400, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
binding.check(L.simdcomp_hip_stream_compress(1, 0, P(ids), nblk, P(strm), P(nb), P(ws_k), st), "compress")
T = int(nb.item())
out = torch.empty_like(ids)
r3 = torch.empty(3, dtype=torch.int64, device=dev)
ws_d = torch.empty(L.simdcomp_hip_stream_decode_workspace(T, nblk), dtype=torch.uint8, device=dev)


def timeit(f, reps=9):
    f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return round(best, 4)


res = {}
for name, call in (("stream_decode", lambda s: L.simdcomp_hip_stream_decode(1, 0, P(strm), T, nblk, P(out), P(r3), P(ws_d), s)),
                   ("stream_compress", lambda s: L.simdcomp_hip_stream_compress(1, 0, P(ids), nblk, P(strm), P(nb), P(ws_k), s))):
    res[name] = {"plain_ms": timeit(lambda: binding.check(call(torch.cuda.current_stream().cuda_stream), name))}
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        binding.check(call(torch.cuda.current_stream().cuda_stream), name)
    res[name]["graph_ms"] = timeit(g.replay)
    torch.cuda.synchronize()
    if name == "stream_decode":
        assert torch.equal(out, ids) and r3.tolist() == [nblk, T - nblk, 0], r3.tolist()
print("graph", json.dumps(res))
