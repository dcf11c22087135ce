import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from simdcomp_amd import binding, codec
dev = torch.device("cuda:0")
n = 1 << 30; nblk = n // 128
w = torch.minimum(torch.randint(0, 21, (nblk,), device=dev), torch.randint(0, 21, (nblk,), device=dev))
gaps = torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device=dev)
gaps &= ((1 << w) - 1).to(torch.int32).repeat_interleave(128)
ids = torch.cumsum(gaps, 0, dtype=torch.int32); del gaps
buf, nbytes = codec.stream_compress(1, ids)
T = int(nbytes.item())
L, P = binding.lib(), binding.ptr
ibits = torch.empty(nblk, dtype=torch.uint8, device=dev); ioff = torch.empty(nblk, dtype=torch.int64, device=dev)
r3 = torch.empty(3, dtype=torch.int64, device=dev)
ws = torch.empty(L.simdcomp_hip_stream_index_workspace(T), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
def t(f, reps=5):
    f(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
print(os.environ.get("SC_CUT"), "index ms", t(lambda: L.simdcomp_hip_stream_index(P(buf), T, nblk, P(ibits), P(ioff), P(r3), P(ws), st)))
