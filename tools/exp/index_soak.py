"""Soak of the record index: many synthetic byte streams of many shapes (natural, width-like payloads, constant
bytes, runs of zero-width records, wide records, bad bytes, truncations), each compared with the CPU walk of the
oracle: record count, status, and widths / offsets of every record in front of the first bad byte.

    python tools/exp/index_soak.py [streams=300] [seed=1]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from simdcomp_amd import codec  # noqa: E402


def make(rng, kind, target):
    choices = {
        "natural": [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 17, 20, 24, 32],
        "narrow": [0, 0, 1, 1, 1, 2, 3],
        "wide": [28, 30, 31, 32, 32],
        "zeros": [0, 0, 0, 0, 0, 0, 0, 1],
        "mixed": [0, 0, 1, 2, 3, 5, 8, 13, 20, 31, 32],
    }
    ws = choices.get(kind, choices["mixed"])
    widths = []
    total = 0
    while True:
        w = int(rng.choice(ws))
        if total + 1 + 16 * w > target:
            break
        widths.append(w)
        total += 1 + 16 * w
    while total < target:
        widths.append(0)
        total += 1
    widths = np.array(widths, np.uint8)
    pos = np.concatenate([[0], np.cumsum(1 + 16 * widths.astype(np.int64))])[:-1]
    fill = rng.integers(0, 5)
    if fill == 0:
        stream = rng.integers(0, 256, target, dtype=np.uint8)
    elif fill == 1:  # half of the payload bytes look like width bytes, many are zero
        stream = rng.integers(0, 256, target, dtype=np.uint8)
        low = rng.random(target) < 0.5
        stream[low] = rng.integers(0, 33, int(low.sum()), dtype=np.uint8)
        stream[rng.random(target) < 0.2] = 0
    elif fill == 2:  # every payload byte is a valid width byte
        stream = rng.integers(0, 33, target, dtype=np.uint8)
    elif fill == 3:  # payload bytes repeat the record's own width: walks of every phase stay alive
        stream = np.repeat(widths, 1 + 16 * widths.astype(np.int64))
    else:
        stream = np.full(target, int(rng.integers(0, 4)), np.uint8)
    stream[pos] = widths
    return stream, widths


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    orc = oracle.Oracle()
    kinds = ["natural", "narrow", "wide", "zeros", "mixed"]
    bad = 0
    for it in range(count):
        kind = kinds[it % len(kinds)]
        target = int(rng.choice([1, 17, 513, 1539, 16384, 16385, 40000, 3 * 16384 - 1, 300000, 1 << 20, (1 << 22) + 77]))
        if rng.random() < 0.5:
            target = int(rng.integers(1, target + 1))
        stream, widths = make(rng, kind, target)
        damage = rng.integers(0, 4)
        if damage == 1 and widths.size > 1:  # a bad width byte somewhere
            pos = np.concatenate([[0], np.cumsum(1 + 16 * widths.astype(np.int64))])[:-1]
            stream[int(pos[int(rng.integers(0, widths.size))])] = int(rng.integers(33, 256))
        elif damage == 2 and target > 1:  # cut short
            stream = stream[: int(rng.integers(1, target))]
        cap = widths.size + 3
        n, wbits, woff, status = orc.stream_walk(stream, cap)
        bits, off, res = codec.stream_index(torch.from_numpy(np.ascontiguousarray(stream)).cuda(), cap)
        res = res.tolist()
        ok = res[2] == (1 if status else 0)
        if status == 0:
            ok = ok and res[0] == n and res[1] == stream.size - n
        good = n if status == 0 else None
        if good is None:
            # records in front of the first bad byte / of the record that does not fit
            p, good = 0, 0
            while p < stream.size and stream[p] <= 32 and p + 1 + 16 * int(stream[p]) <= stream.size:
                p += 1 + 16 * int(stream[p])
                good += 1
        good = min(good, cap)
        gb, go = bits.cpu().numpy()[:good], off.cpu().numpy().view(np.uint64)[:good]
        p = np.concatenate([[0], np.cumsum(1 + 16 * gb.astype(np.int64))])[:-1] if good else np.zeros(0, np.int64)
        ok = ok and np.array_equal(gb, stream[p] if good else gb) and np.array_equal(go.astype(np.int64), p - np.arange(good))
        if not ok:
            bad += 1
            print("MISMATCH", it, kind, target, stream.size, damage, res, (n, status), flush=True)
        if it % 25 == 24:
            print(f"{it + 1} streams, {bad} mismatches", flush=True)
    print("soak done:", count, "streams,", bad, "mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
