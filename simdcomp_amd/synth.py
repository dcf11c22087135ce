"""Counter-based synthetic integer data (numpy, host side).

Every value is a pure function of (seed, index), so any slice can be generated
independently and the same arrays can be rebuilt on another machine without
shipping them.  The distributions follow the reference's own benchmark
generators (benchmarks/bitpackingbenchmark.c:34-53: values masked to b bits, and
running sums of such values for the differential case).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """splitmix64 finaliser applied elementwise to a uint64 array."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def rand_u32(seed, n, start=0):
    """n pseudo-random uint32 values h(seed, start + i)."""
    idx = np.arange(start, start + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = idx + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)
    return (splitmix64(key) & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def mask(bit):
    return np.uint32(0xFFFFFFFF if bit >= 32 else (1 << bit) - 1)


def random_bits(seed, n, bit, start=0):
    """Random-order values below 2**bit (bitpackingbenchmark.c:34-42)."""
    return rand_u32(seed, n, start) & mask(bit)


def sorted_gaps(seed, n, bit, first=0):
    """Non-decreasing (mod 2**32) sequence whose successive differences are
    random values below 2**bit (bitpackingbenchmark.c:44-53)."""
    gaps = random_bits(seed, n, bit).astype(np.uint64)
    return ((np.cumsum(gaps) + np.uint64(first)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def block_offsets(values, initvalue=0):
    """Per-block differential seeds for a chained stream: block k starts from the
    last value of block k-1 (example.c:63-71), block 0 from initvalue."""
    nblocks = values.size // 128
    off = np.empty(nblocks, dtype=np.uint32)
    if nblocks:
        off[0] = initvalue
        off[1:] = values[127:(nblocks - 1) * 128:128]
    return off


def clustered(seed, nblocks, bits):
    """Frame-of-reference corpus: block k holds base[k] + r, r < 2**bits[k]
    (wrapping mod 2**32).  Returns (values, base)."""
    bits = np.broadcast_to(np.asarray(bits, dtype=np.uint8), (nblocks,))
    base = rand_u32(seed, nblocks)
    r = rand_u32(seed + 1, nblocks * 128).reshape(nblocks, 128)
    m = np.where(bits >= 32, np.uint64(0xFFFFFFFF), (np.uint64(1) << bits.astype(np.uint64)) - np.uint64(1))
    r = r & m.astype(np.uint32)[:, None]
    with np.errstate(over="ignore"):
        vals = (base[:, None] + r).astype(np.uint32)
    return vals.reshape(-1), base


def skewed_widths(seed, nblocks, lo=1, hi=20):
    """Per-block widths in [lo, hi], skewed towards small values (minimum of two
    uniform draws)."""
    a = rand_u32(seed, nblocks) % np.uint32(hi - lo + 1)
    b = rand_u32(seed + 7, nblocks) % np.uint32(hi - lo + 1)
    return (np.minimum(a, b) + np.uint32(lo)).astype(np.uint8)
