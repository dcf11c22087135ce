"""Development probe: the arbitrary-length entry points at 2^30 - 77 integers."""
import sys, json, torch
sys.path.insert(0, '.')
from simdcomp_amd import codec
n = (1 << 30) - 77; dev = torch.device("cuda:0")
def t(f, reps=5):
    f(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
x = torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device=dev)
out = torch.empty(n, dtype=torch.int32, device=dev)
for b in (4, 8, 16, 24):
    xs = x & ((1 << b) - 1)
    nb = codec.compressedbytes(n, b)
    pay = torch.empty(nb + 64, dtype=torch.uint8, device=dev)
    res = {}
    for name, f in (("pack_length", lambda: codec.pack_length(xs, b, out=pay)),
                    ("unpack_length", lambda: codec.unpack_length(pay, n, b, out=out)),
                    ("packFOR_length", lambda: codec.packFOR_length(0, xs, b, out=pay)),
                    ("unpackFOR_length", lambda: codec.unpackFOR_length(0, pay, n, b, out=out))):
        ms = t(f)
        res[name] = {"ms": round(ms, 4), "frac": round((4 * n + nb) / ms / 1e6 / 8000, 3)}
    print(b, json.dumps(res), flush=True)
