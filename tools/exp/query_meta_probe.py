"""Development probe: what do the three metadata fetches of a random query cost?  The same 2^24 random (block, slot /
key) queries on blocks of one width b, addressed (a) with per-block widths, offsets and seeds (three scattered
metadata reads per query), (b) as a uniform-width array with seeds (one), (c) uniform with one seed for all (none)."""
import sys, torch
sys.path.insert(0, '.')
from simdcomp_amd import codec
dev = torch.device("cuda:0")
nblk = 1 << 23
nq = 1 << 24
def t(f, reps=5):
    f(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); e1.synchronize(); best = min(best, e0.elapsed_time(e1))
    return round(best, 4)
for b in (4, 7, 12):
    pay = torch.randint(0, 256, (nblk * 16 * b,), dtype=torch.uint8, device=dev)
    bits = torch.full((nblk,), b, dtype=torch.uint8, device=dev)
    off, _ = codec.payload_offsets(bits)
    seeds = torch.randint(0, 1 << 20, (nblk,), dtype=torch.int32, device=dev)
    qb = torch.randint(0, nblk, (nq,), device=dev)
    slots = torch.randint(0, 128, (nq,), dtype=torch.int32, device=dev)
    keys = torch.randint(0, 1 << 24, (nq,), dtype=torch.int32, device=dev)
    row = {"b": b}
    for name, kw in (("three_arrays", dict(bits=bits, offsets=off, initvalues=seeds)), ("seeds_only", dict(bit=b, initvalues=seeds)),
                     ("no_metadata", dict(bit=b, initvalue=5))):
        row["select_" + name] = t(lambda: codec.select(1, pay, slots, query_blocks=qb, **kw))
        row["search_" + name] = t(lambda: codec.search(1, pay, keys, query_blocks=qb, **kw))
    print(row, flush=True)
    del pay
