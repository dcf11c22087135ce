#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/...) into the small files kept under profiles/.

    python tools/summarize_profiles.py ROUND STATS_CSV FETCH_CSV WRITE_CSV

Writes profiles/rROUND_kernel_stats.csv (the decode/encode kernels' rows of the
--kernel-trace --stats summary), profiles/rROUND_pmc_traffic.json (per width:
FETCH_SIZE, WRITE_SIZE and the HBM bytes derived from them) and refreshes
profiles/pmc_traffic_latest.json, which bench.py reads for roofline.traffic.

HBM bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: on gfx950 FETCH_SIZE counts
128-byte read requests as 64 bytes for wide coalesced streams (the b = 32 copy
kernel calibrates it: 2 x FETCH_SIZE equals the bytes it must read), WRITE_SIZE
is exact for 16-byte-per-lane stores.  The two counters come from separate
--pmc passes.
"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_width(path, pattern):
    d = collections.defaultdict(list)
    nblk = None
    for r in csv.DictReader(open(path)):
        m = re.search(pattern, r["Kernel_Name"])
        if m:
            d[int(m.group(1))].append(float(r["Counter_Value"]))
            nblk = int(r["Grid_Size"])
    return d


def main():
    rnd, stats, fetch, write = sys.argv[1:5]
    out = os.path.join(ROOT, "profiles")
    rows = list(csv.DictReader(open(stats)))
    keep = [r for r in rows if "sc::k_" in r["Name"]]
    with open(os.path.join(out, f"r{rnd}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)
    pat = r"k_unpack_fast<(\d+), 1, \d+, \d+, \d+(?:, \d+)?>"
    fd, wd = per_width(fetch, pat), per_width(write, pat)
    table = {}
    for b in sorted(fd):
        fkb, wkb = fd[b][-1], wd[b][-1]
        table[str(b)] = {"FETCH_SIZE_KB": fkb, "WRITE_SIZE_KB": wkb, "hbm_bytes": int(2 * fkb * 1024 + wkb * 1024)}
    doc = {"kernel": "sc::k_unpack_fast<b, d1>", "launch": "2^23 blocks (2^30 integers) per launch",
           "correction": "read bytes = 2 * FETCH_SIZE * 1024 (gfx950 counts 128-B requests as 64 B); write bytes = WRITE_SIZE * 1024",
           "per_width": table}
    for name in (f"r{rnd}_pmc_traffic.json", "pmc_traffic_latest.json"):
        with open(os.path.join(out, name), "w") as f:
            json.dump(doc, f, indent=1)
    print(json.dumps(table)[:400])


if __name__ == "__main__":
    main()
