#!/usr/bin/env python3
"""Summary of tools/exp/state_counters.sh: per counter pass and allocation mode, the fraction of the HBM peak each
width ran at in that very process (its state) next to the pass's counters per launch of that width's decode kernel.

    python tools/exp/state_summary.py gpurun_out/r03/state profiles/r03_state_counters.json [widths...]"""
import collections
import csv
import glob
import json
import os
import re
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    widths = [int(w) for w in sys.argv[3:]] or [1, 4, 8, 10, 12, 14, 20, 24, 28, 32]
    pat = re.compile(r"k_unpack_fast<(\d+), 1, ")
    doc = {"what": "bench.py decode steps under rocprofv3 --pmc (tools/exp/state_counters.sh): per pass and allocation mode "
                   "(0: an allocation per array, 2: one allocation for the working set) the fraction of the HBM peak measured "
                   "in that process (HIP events; under the profiler) and the counters per launch, summed over all instances",
           "passes": {}, "plain": {}}
    for line in sorted(glob.glob(os.path.join(src, "*.json"))):
        name = os.path.basename(line)[:-5]
        try:
            d = json.loads(open(line).read().strip().splitlines()[-1])
        except Exception:
            continue
        fr = {str(b): d["roofline"]["per_width"][str(b)]["frac"] for b in widths}
        row = {"aggregate_frac": d["roofline"]["frac"], "frac": fr}
        if name.startswith("plain"):
            doc["plain"][name] = row
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(set)
        for path in glob.glob(os.path.join(src, name, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                m = pat.search(r["Kernel_Name"])
                if m and int(m.group(1)) in widths:
                    b = int(m.group(1))
                    per[b][r["Counter_Name"]] += float(r["Counter_Value"])
                    cnt[b].add(r["Dispatch_Id"])
        row["counters_per_launch"] = {str(b): {c: v / max(1, len(cnt[b])) for c, v in per[b].items()} for b in widths if b in per}
        doc["passes"][name] = row
    json.dump(doc, open(out, "w"), indent=1)
    # side by side on the terminal: for each pass, mode 0 against mode 2 at every width
    for p in sorted(set(k[:2] for k in doc["passes"])):
        a, b = doc["passes"].get(p + "_m0"), doc["passes"].get(p + "_m2")
        if not a or not b:
            continue
        print(f"== {p}: aggregate {a['aggregate_frac']} (mode 0) / {b['aggregate_frac']} (mode 2)")
        for w in widths:
            ca, cb = a["counters_per_launch"].get(str(w), {}), b["counters_per_launch"].get(str(w), {})
            cells = "  ".join(f"{c.replace('_sum', '')}: {ca[c]:.4g} / {cb.get(c, float('nan')):.4g}" for c in sorted(ca))
            print(f"  b={w:2d} frac {a['frac'][str(w)]:.3f} / {b['frac'][str(w)]:.3f}   {cells}")


if __name__ == "__main__":
    main()
