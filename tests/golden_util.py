"""Loader for the committed golden fixtures (tests/golden/*.npz, written by
tools/gen_golden.py from the compiled reference) and the input recipes that go
with them.  Data only; numpy.load without pickle."""
import os

import numpy as np

from simdcomp_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FULL_LENGTHS = [0, 1, 2, 3, 4, 5, 7, 8, 31, 32, 33, 127, 128, 129, 130, 255, 257]
DIGEST_LENGTHS = [1000, 4095, 9999]
PACK_NAMES = ["simdpack", "simdpackwithoutmask", "simdpackd1", "simdpackwithoutmaskd1", "simdpackFOR"]


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def length_input(n, bit, kind):
    if kind == 0:
        return synth.random_bits(500 + bit, n, bit)
    return synth.rand_u32(600 + bit, n)


def corpus_plain(bit, masked, nblk):
    return synth.random_bits(700, nblk * 128, bit) if not masked else synth.rand_u32(701, nblk * 128)


def corpus_d1(bit, nblk):
    v = synth.sorted_gaps(710 + bit, nblk * 128, bit, first=99)
    return v, synth.block_offsets(v, 17)


def corpus_for(bit, nblk):
    return synth.clustered(720 + bit, nblk, bit)


def corpus_mixed(nblk):
    widths = synth.skewed_widths(730, nblk, 1, 20)
    gaps = synth.rand_u32(731, nblk * 128).reshape(nblk, 128)
    m = ((np.uint64(1) << widths.astype(np.uint64)) - np.uint64(1)).astype(np.uint32)
    gaps = (gaps & m[:, None]).reshape(-1).astype(np.uint64)
    v = (np.cumsum(gaps) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    return v, synth.block_offsets(v, 0)


def stream_corpus(seed, nblk):
    """Integers for the example.c byte-stream tests: mostly sorted with per-block
    gap widths 1..20, plus runs of equal values (0-bit blocks, which pass the
    differential offset through) and blocks of arbitrary 32-bit values (32-bit
    blocks, which restart it)."""
    widths = synth.skewed_widths(seed, nblk, 1, 20)
    kind = synth.rand_u32(seed + 1, nblk) % np.uint32(16)
    gaps = synth.rand_u32(seed + 2, nblk * 128).reshape(nblk, 128)
    m = ((np.uint64(1) << widths.astype(np.uint64)) - np.uint64(1)).astype(np.uint32)
    gaps = gaps & m[:, None]
    gaps[kind == 0] = 0
    v = (np.cumsum(gaps.reshape(-1).astype(np.uint64)) & np.uint64(0xFFFFFFFF)).astype(np.uint32).reshape(nblk, 128)
    wild = synth.rand_u32(seed + 3, nblk * 128).reshape(nblk, 128) | np.uint32(0x80000000)
    wild[:, 1::2] &= np.uint32(0x7FFFFFFF)  # alternate high/low so that successive differences need all 32 bits
    v[kind == 1] = wild[kind == 1]
    return np.ascontiguousarray(v.reshape(-1))


QUERY_LENGTHS = [128, 127, 65, 4, 3, 1]
QUERY_SLOTS = [0, 1, 2, 3, 4, 63, 64, 126, 127, 133]


def query_block(bit, kind):
    """One block for the select / search fixtures: kind 0 sorted with gaps below 2^bit,
    kind 1 arbitrary 32-bit values (unsorted: exercises what the reference's searches do then)."""
    init = 990 + bit
    if bit == 0:
        return np.full(128, init, np.uint32), init
    if kind == 0:
        return synth.sorted_gaps(800 + bit, 128, min(bit, 25), first=init), init
    return synth.rand_u32(850 + bit, 128), init


def query_keys(dec, bit, kind):
    extra = synth.rand_u32(870 + 2 * bit + kind, 4)
    ks = [0, 1, int(dec[0]), int(dec[5]), int(dec[64]) + 1, int(dec[127]), int(dec[127]) + 1, 0xFFFFFFFF, 0x80000000]
    return np.array([k & 0xFFFFFFFF for k in ks] + [int(e) for e in extra], dtype=np.uint32)
