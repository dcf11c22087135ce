#!/usr/bin/env python3
"""Merge the counter_collection.csv files of several rocprofv3 --pmc runs of tools/pmc_workload.py into one JSON:
per kernel (template arguments kept, argument lists dropped) the average of every counter over the kernel's
dispatches, plus derived figures where their inputs are present:

  hbm_bytes        2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (gfx950 tallies 128-byte read requests as 64 bytes)
  valu_busy        SQ_ACTIVE_INST_VALU * 4 / (SIMDs * GRBM_GUI_ACTIVE / 8)  share of the kernel's cycles a SIMD's
                   vector ALU is busy (1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs)
  lds_wait_share   SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES   share of the wave-cycles spent waiting on an LDS instruction
  waves_per_simd   SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / 4 ... average resident waves per SIMD while the SQ is busy

    python tools/pmc_summary.py OUT.json DIR [DIR...]"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)  # drop the argument list
    return name.replace("void ", "").strip()


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            per_dispatch = collections.defaultdict(float)
            names = {}
            for r in csv.DictReader(open(path)):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = r["Kernel_Name"]
            for (disp, ctr), v in per_dispatch.items():
                k = names[disp]
                if "sc::" in k:
                    acc[short(k)][ctr].append(v)
    doc = {}
    for k in sorted(acc):
        row = {c: sum(v) / len(v) for c, v in acc[k].items()}
        row["dispatches"] = max(len(v) for v in acc[k].values())
        if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
            row["hbm_bytes"] = int(2 * row["FETCH_SIZE"] * 1024 + row["WRITE_SIZE"] * 1024)
        if "SQ_ACTIVE_INST_VALU" in row and row.get("GRBM_GUI_ACTIVE"):
            row["valu_busy"] = round(row["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * row["GRBM_GUI_ACTIVE"] / 8), 3)
        if "SQ_WAIT_INST_LDS" in row and row.get("SQ_WAVE_CYCLES"):
            row["lds_wait_share"] = round(row["SQ_WAIT_INST_LDS"] / row["SQ_WAVE_CYCLES"], 3)
        if row.get("SQ_BUSY_CYCLES") and "SQ_WAVE_CYCLES" in row:
            row["waves_per_simd"] = round(row["SQ_WAVE_CYCLES"] / row["SQ_BUSY_CYCLES"] / 4, 2)
        doc[k] = row
    json.dump(doc, open(out, "w"), indent=1)
    for k, row in doc.items():
        print(k[:70], {a: b for a, b in row.items() if a in ("hbm_bytes", "valu_busy", "lds_wait_share", "waves_per_simd")})


if __name__ == "__main__":
    main()
