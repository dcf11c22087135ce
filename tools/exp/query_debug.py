import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import golden_util as G
import oracle
from simdcomp_amd import codec as dev, synth
orc = oracle.Oracle()
def to_dev(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32: return torch.from_numpy(a.view(np.int32)).cuda()
    if a.dtype == np.uint64: return torch.from_numpy(a.view(np.int64)).cuda()
    return torch.from_numpy(a).cuda()
u32 = lambda t: t.cpu().numpy().view(np.uint32)
nblk, nq = 3000, 20011
x = G.stream_corpus(61, nblk)
d_x = to_dev(x)
seeds = torch.empty(nblk, dtype=torch.int32, device="cuda")
bits = dev.maxbitsd1(d_x, initvalue=0, initvalues_out=seeds)
off, total = dev.payload_offsets(bits)
arena = torch.zeros(int(total.item()) + 16, dtype=torch.uint8, device="cuda")
dev.pack_mixed(1, d_x, bits, off, arena, initvalues=seeds)
qb = (synth.rand_u32(1, nq) % np.uint32(nblk)).astype(np.uint64)
slots = (synth.rand_u32(2, nq) % np.uint32(256)).astype(np.int32)
want = x.reshape(nblk, 128)[qb.astype(np.int64), slots & 127]
got = u32(dev.select(1, arena, to_dev(slots), bits=bits, offsets=off, initvalues=seeds, query_blocks=to_dev(qb)))
bad = np.nonzero(got != want)[0]
hb = bits.cpu().numpy()
print("mismatches", bad.size, "of", nq)
for q in bad[:40]:
    print(q, q % 64, "blk", qb[q], "b", hb[int(qb[q])], "slot", slots[q] & 127, "got", got[q], "want", want[q],
          "prev blk", qb[q - 1] if q else None, "next", qb[q + 1] if q + 1 < nq else None)
