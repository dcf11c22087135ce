"""How many slots does the record index use on the microbenchmark's stream, and how many went without a log?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
wild = torch.randint(0, nblk, (nblk // 50,), device=dev)
ids.view(-1, 128)[wild] = torch.randint(-2 ** 31, 2 ** 31 - 1, (wild.numel(), 128), dtype=torch.int32, device=dev)
seeds = torch.empty(nblk, dtype=torch.int32, device=dev)
bits = codec.maxbitsd1(ids, initvalue=0, initvalues_out=seeds)
off, total = codec.payload_offsets(bits)
T = int(total.item())
arena = torch.empty(T + 16, dtype=torch.uint8, device=dev)
codec.pack_mixed(1, ids, bits, off, arena, initvalues=seeds)
strm = codec.stream_build(arena, bits, off, T)
L, P = binding.lib(), binding.ptr
nbytes = T + nblk
wsb = L.simdcomp_hip_stream_index_workspace(nbytes)
ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
ibits = torch.empty(nblk, dtype=torch.uint8, device=dev)
ioff = torch.empty(nblk, dtype=torch.int64, device=dev)
r3 = torch.empty(3, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
L.simdcomp_hip_stream_index(P(strm), nbytes, nblk, P(ibits), P(ioff), P(r3), P(ws), st)
torch.cuda.synchronize()
a256 = lambda x: (x + 255) // 256 * 256
nchunks = (nbytes + 16383) // 16384
ngroups = (nchunks + 31) // 32
nsupers = (ngroups + 31) // 32
SLOTS = 4
at = a256(4 * 513 * nchunks) + a256(8 * 513 * ngroups) + a256(8 * 513 * nsupers) + a256(8 * nsupers) + a256(8 * ngroups) + a256(8 * nchunks)
sl = a256(4 * SLOTS * nchunks)
start = ws[at:at + 4 * SLOTS * nchunks].view(torch.int32).view(nchunks, SLOTS)
result = ws[at + sl:at + sl + 4 * SLOTS * nchunks].view(torch.int32).view(nchunks, SLOTS)
log = ws[at + 2 * sl:at + 2 * sl + 4 * SLOTS * nchunks].view(torch.int32).view(nchunks, SLOTS)
pool = ws[at + 3 * sl:at + 3 * sl + 4].view(torch.int32).item()
active = start != -1
print("chunks", nchunks, "active slots", int(active.sum()), "per chunk", float(active.sum()) / nchunks, "regions", nchunks + nchunks // 2 + 64, "pool counter", pool)
print("slots per chunk histogram", torch.bincount(active.sum(1)).tolist())
print("active slots without a log", int((active & (log == -1)).sum()), " dead slots (bad byte)", int((active & (result == -1)).sum()))
print(r3.tolist())
