#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref).

Run in the container that holds /root/reference, after `make -C oracle`:

    python tools/gen_golden.py

The fixtures are data only (inputs, parameters, expected outputs, digests).
They pin the scalar restatement and the HIP path on machines where neither the
reference tree nor its compiled form exists.

Files written
  blocks.npz    one-block vectors for b = 0..32 on in-range, arbitrary 32-bit
                and sorted inputs: every pack flavour, the three unpackers on
                the packed bytes, and the per-block reductions
  lengths.npz   simdpack_length / simdpackFOR_length outputs (and their
                decodes) for a list of lengths x all widths, small cases in
                full and FNV-1a-64 digests for the long ones
  corpus.npz    FNV-1a-64 digests of packed payload and decoded output for
                seeded multi-thousand-block corpora (plain / d1 / FOR, uniform
                widths and a mixed-width d1 stream)
  stream.npz    the [width byte][payload] byte streams of example.c:56-75 (d1)
                and :135-150 (plain) over 600 blocks that include 0-bit and
                32-bit blocks, written by the reference's block routines
  queries.npz   simdselectd1 / simdsearchwithlengthd1 / simdsearchd1 / simdscand1
                and simdselectFOR / simdsearchwithlengthFOR answers of the
                reference on sorted and unsorted blocks of every width
  lanes.npz     avxpack / avx512pack families (8- and 16-lane wire formats) on
                one block per width (needs a CPU with AVX2 and AVX-512F)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from simdcomp_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FULL_LENGTHS = [0, 1, 2, 3, 4, 5, 7, 8, 31, 32, 33, 127, 128, 129, 130, 255, 257]
DIGEST_LENGTHS = [1000, 4095, 9999]
CORPUS_BLOCKS = 4096


def fnv(a):
    h = 0xCBF29CE484222325
    # vectorised FNV is not possible; use the C helper from the restatement build
    return ORC.fnv1a64(a)


def block_inputs(bit, kind):
    if kind == 0:
        return synth.random_bits(100 + bit, 128, bit)
    if kind == 1:
        return synth.rand_u32(200 + bit, 128)
    return synth.sorted_gaps(300 + bit, 128, min(bit, 25), first=0x1234 * (bit + 1))


def gen_blocks(ref):
    nb, nk = 33, 3
    x = np.zeros((nb, nk, 128), np.uint32)
    init = np.zeros((nb, nk), np.uint32)
    packed = np.zeros((nb, nk, 5, 512), np.uint8)  # pack, packwithoutmask, packd1, packwithoutmaskd1, packFOR
    unpacked = np.zeros((nb, nk, 5, 3, 128), np.uint32)  # x {unpack, unpackd1, unpackFOR}
    red = np.zeros((nb, nk, 4), np.uint32)  # maxbits, maxbitsd1, min, max
    for bit in range(nb):
        for kind in range(nk):
            v = block_inputs(bit, kind)
            iv = int(synth.rand_u32(400 + bit, 3)[kind]) if kind != 0 else (bit * 3)
            x[bit, kind] = v
            init[bit, kind] = iv
            outs = [ref.simdpack(v, bit), ref.simdpackwithoutmask(v, bit), ref.simdpackd1(iv, v, bit),
                    ref.simdpackwithoutmaskd1(iv, v, bit), ref.simdpackFOR(iv, v, bit)]
            for j, o in enumerate(outs):
                packed[bit, kind, j, :16 * bit] = o
                unpacked[bit, kind, j, 0] = ref.simdunpack(o, bit)
                unpacked[bit, kind, j, 1] = ref.simdunpackd1(iv, o, bit)
                unpacked[bit, kind, j, 2] = ref.simdunpackFOR(iv, o, bit)
            lo, hi = ref.simdmaxmin(v)
            red[bit, kind] = [ref.maxbits(v), ref.simdmaxbitsd1(iv, v), lo, hi]
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), x=x, init=init, packed=packed, unpacked=unpacked,
                        red=red)


def length_input(n, bit, kind):
    if kind == 0:
        return synth.random_bits(500 + bit, n, bit)
    return synth.rand_u32(600 + bit, n)


def gen_lengths(ref):
    full = {}
    rows = []  # (n, bit, kind, nbytes, fnv(pack_length), fnv(packFOR_length), fnv(unpack), fnv(unpackFOR))
    base = 0x00C0FFEE
    for bit in range(33):
        for kind in range(2):
            for n in FULL_LENGTHS + DIGEST_LENGTHS:
                v = length_input(n, bit, kind)
                p = ref.simdpack_length(v, bit)
                pf = ref.simdpackFOR_length(base, v, bit)
                u, used = ref.simdunpack_length(p, n, bit)
                uf, usedf = ref.simdunpackFOR_length(base, pf, n, bit)
                assert used == len(p) == ref.compressedbytes(n, bit)
                assert usedf == len(pf) == ref.compressedbytesFOR(n, bit)
                rows.append((n, bit, kind, len(p), fnv(p), fnv(pf), fnv(u), fnv(uf)))
                if n in FULL_LENGTHS:
                    key = f"n{n}_b{bit}_k{kind}"
                    full[key + "_p"] = p
                    full[key + "_pf"] = pf
                    full[key + "_u"] = u
                    full[key + "_uf"] = uf
    table = np.array(rows, dtype=np.uint64)
    np.savez_compressed(os.path.join(OUT, "lengths.npz"), table=table, base=np.uint32(base), **full)


def gen_corpus(ref):
    """Digests of block-by-block reference output over seeded corpora."""
    nblk = CORPUS_BLOCKS
    rows = []  # (mode, bit, masked, fnv(payload), fnv(decoded))
    for bit in range(33):
        # plain: random order, in range for the unmasked packer, arbitrary for the masked one
        for masked in (0, 1):
            v = synth.random_bits(700, nblk * 128, bit) if not masked else synth.rand_u32(701, nblk * 128)
            f = ref.simdpack if masked else ref.simdpackwithoutmask
            pay = np.concatenate([f(v[k * 128:(k + 1) * 128], bit) for k in range(nblk)]) if bit else \
                np.zeros(0, np.uint8)
            dec = np.concatenate([ref.simdunpack(pay[k * 16 * bit:(k + 1) * 16 * bit], bit) for k in range(nblk)])
            rows.append((0, bit, masked, fnv(pay), fnv(dec)))
        # d1: sorted, chained offsets
        v = synth.sorted_gaps(710 + bit, nblk * 128, bit, first=99)
        offs = synth.block_offsets(v, 17)
        for masked in (0, 1):
            f = ref.simdpackd1 if masked else ref.simdpackwithoutmaskd1
            pay = np.concatenate([f(int(offs[k]), v[k * 128:(k + 1) * 128], bit) for k in range(nblk)]) if bit \
                else np.zeros(0, np.uint8)
            dec = np.concatenate([ref.simdunpackd1(int(offs[k]), pay[k * 16 * bit:(k + 1) * 16 * bit], bit)
                                  for k in range(nblk)])
            rows.append((1, bit, masked, fnv(pay), fnv(dec)))
        # FOR: clustered
        v, base = synth.clustered(720 + bit, nblk, bit)
        pay = np.concatenate([ref.simdpackFOR(int(base[k]), v[k * 128:(k + 1) * 128], bit) for k in range(nblk)]) \
            if bit else np.zeros(0, np.uint8)
        dec = np.concatenate([ref.simdunpackFOR(int(base[k]), pay[k * 16 * bit:(k + 1) * 16 * bit], bit)
                              for k in range(nblk)])
        rows.append((2, bit, 0, fnv(pay), fnv(dec)))
    # mixed-width chained d1 stream, the example.c:56-75 recipe with widths from simdmaxbitsd1
    widths = synth.skewed_widths(730, nblk, 1, 20)
    gaps = synth.rand_u32(731, nblk * 128).reshape(nblk, 128)
    m = ((np.uint64(1) << widths.astype(np.uint64)) - np.uint64(1)).astype(np.uint32)
    gaps = (gaps & m[:, None]).reshape(-1).astype(np.uint64)
    v = (np.cumsum(gaps) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    offs = synth.block_offsets(v, 0)
    bits = np.array([ref.simdmaxbitsd1(int(offs[k]), v[k * 128:(k + 1) * 128]) for k in range(nblk)], np.uint8)
    pay = np.concatenate([ref.simdpackwithoutmaskd1(int(offs[k]), v[k * 128:(k + 1) * 128], int(bits[k]))
                          for k in range(nblk)])
    pos = np.concatenate([[0], np.cumsum(bits.astype(np.uint64) * 16)]).astype(np.uint64)
    dec = np.concatenate([ref.simdunpackd1(int(offs[k]), pay[int(pos[k]):int(pos[k + 1])], int(bits[k]))
                          for k in range(nblk)])
    assert np.array_equal(dec, v)
    mixed = np.array([fnv(bits), fnv(pay), fnv(dec)], dtype=np.uint64)
    np.savez_compressed(os.path.join(OUT, "corpus.npz"), table=np.array(rows, dtype=np.uint64), mixed=mixed,
                        nblocks=np.uint64(nblk))


def gen_stream(ref):
    """The example.c byte streams produced by the reference's own block routines."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_util

    nblk = 600
    x = golden_util.stream_corpus(900, nblk)
    d1 = ORC.stream_compress(x, d1=True, initvalue=0, codec=ref)
    plain = ORC.stream_compress(x, d1=False, codec=ref)
    assert np.array_equal(ORC.stream_decode(d1, nblk, d1=True, codec=ref), x)
    assert np.array_equal(ORC.stream_decode(plain, nblk, d1=False, codec=ref), x)
    np.savez_compressed(os.path.join(OUT, "stream.npz"), nblocks=np.uint64(nblk), d1=d1, plain=plain)


def gen_queries(ref):
    """simdselectd1 / simdsearchwithlengthd1 / simdsearchd1 / simdscand1 and the FOR select / search of the
    reference on one block per width and kind."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_util as G

    nk, nl, ns = 13, len(G.QUERY_LENGTHS), len(G.QUERY_SLOTS)
    sel = np.zeros((33, 2, 2, ns), np.uint32)            # [bit][kind][d1, FOR][slot]
    srch = np.zeros((33, 2, 2, nk, nl, 2), np.int64)      # [bit][kind][d1, FOR][key][length](pos, value)
    full = np.zeros((33, 2, nk, 6), np.int64)             # simdsearchd1: pos, value, offset vector after
    scan = np.zeros((33, 2, 4), np.uint32)
    for bit in range(33):
        for kind in range(2):
            x, init = G.query_block(bit, kind)
            d1 = ref.simdpackwithoutmaskd1(init, x, bit)
            fr = ref.simdpackFOR(init, x, bit)
            dec = ref.simdunpackd1(init, d1, bit)
            keys = G.query_keys(dec, bit, kind)
            for i, slot in enumerate(G.QUERY_SLOTS):
                sel[bit, kind, 0, i] = ref.simdselectd1(init, d1, bit, slot)
                sel[bit, kind, 1, i] = ref.simdselectFOR(init, fr, bit, slot & 127)
            for ki, key in enumerate(keys):
                for li, length in enumerate(G.QUERY_LENGTHS):
                    srch[bit, kind, 0, ki, li] = ref.simdsearchwithlengthd1(init, d1, bit, length, int(key))
                    # the reference reads slot `length` when nothing is found: stay inside the block
                    srch[bit, kind, 1, ki, li] = ref.simdsearchwithlengthFOR(init, fr, bit, min(length, 127), int(key))
                pos, val, o4 = ref.simdsearchd1(np.full(4, init, np.uint32), d1, bit, int(key))
                full[bit, kind, ki] = [pos, val] + [int(v) for v in o4]
            scan[bit, kind] = ref.simdscand1(np.full(4, init, np.uint32), d1, bit)
    np.savez_compressed(os.path.join(OUT, "queries.npz"), select=sel, search=srch, searchd1=full, scan=scan)


def gen_lanes():
    """avxpack / avxpackwithoutmask / avxunpack / avxmaxbits and the avx512 family of the reference on one block
    per width: in-range input through the unmasked packer, arbitrary 32-bit input through both."""
    ra = oracle.ReferenceAvx()
    out = {}
    for lanes in (8, 16):
        n = 32 * lanes
        packed = np.zeros((33, 3, 4 * lanes * 32), np.uint8)   # [bit][in-range unmasked, wild masked, wild unmasked]
        unpacked = np.zeros((33, 3, n), np.uint32)
        mb = np.zeros((33, 2), np.uint32)
        for bit in range(33):
            xs = (synth.random_bits(1100 + bit, n, bit), synth.rand_u32(1200 + bit, n))
            for j, (x, masked) in enumerate(((xs[0], False), (xs[1], True), (xs[1], False))):
                c = ra.lanes_pack(lanes, x, bit, masked)
                packed[bit, j, :c.size] = c
                unpacked[bit, j] = ra.lanes_unpack(lanes, c, bit)
            mb[bit] = [ra.lanes_maxbits(lanes, xs[0]), ra.lanes_maxbits(lanes, xs[1])]
        out[f"packed{lanes}"], out[f"unpacked{lanes}"], out[f"maxbits{lanes}"] = packed, unpacked, mb
    np.savez_compressed(os.path.join(OUT, "lanes.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    ORC = oracle.Oracle()  # used only for its FNV helper
    REF = oracle.Reference()
    gen_blocks(REF)
    gen_lengths(REF)
    gen_corpus(REF)
    gen_stream(REF)
    gen_queries(REF)
    if oracle.have_ref_avx():
        gen_lanes()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
