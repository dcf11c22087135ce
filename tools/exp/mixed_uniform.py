"""Development probe: the per-block-width decoder on data whose widths are all equal, against the uniform decoder."""
import sys, torch
sys.path.insert(0, '.')
from simdcomp_amd import codec
n = 1 << 30; nblk = n // 128; dev = torch.device("cuda:0")
def t(f, reps=7):
    f(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
out = torch.empty(n, dtype=torch.int32, device=dev)
seeds = torch.zeros(nblk, dtype=torch.int32, device=dev)
for b in ([int(v) for v in sys.argv[1:]] or (4, 7, 12, 16)):
    pay = torch.randint(0, 256, (nblk * 16 * b,), dtype=torch.uint8, device=dev)
    bits = torch.full((nblk,), b, dtype=torch.uint8, device=dev)
    off, _ = codec.payload_offsets(bits)
    tu = t(lambda: codec.unpackd1(seeds, pay, nblk, b, out=out))
    tm = t(lambda: codec.unpack_mixed(1, pay, bits, off, nblk, initvalues=seeds, out=out))
    # two widths alternating in runs of 8 blocks / of 1 block, same average
    res = [b, round(tu, 4), round(tm, 4)]
    for run in (64, 8, 1):
        idx = (torch.arange(nblk, device=dev) // run) & 1
        bits2 = (b - 2 + 4 * idx).to(torch.uint8)
        off2, tot = codec.payload_offsets(bits2)
        pay2 = torch.randint(0, 256, (int(tot.item()),), dtype=torch.uint8, device=dev)
        res.append(round(t(lambda: codec.unpack_mixed(1, pay2, bits2, off2, nblk, initvalues=seeds, out=out)), 4))
        del pay2
    print("b, uniform ms, mixed-kernel ms on uniform widths, widths b-2/b+2 in runs of 64, 8, 1:", res, flush=True)
    del pay
