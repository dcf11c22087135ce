#!/usr/bin/env python3
"""Merge the counter_collection.csv files of several rocprofv3 --pmc runs of tools/pmc_workload.py into one JSON:
per kernel (template arguments kept, argument lists dropped) the average of every counter over the kernel's
dispatches, plus derived figures where their inputs are present:

  hbm_bytes        2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (gfx950 tallies 128-byte read requests as 64 bytes)
  valu_load        SQ_INSTS_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8): vector-ALU cycles at four per wave
                   instruction over the cycles the kernel ran (GRBM_GUI_ACTIVE is summed over the 8 XCDs).  A relative
                   figure: kernels that are bound by vector-ALU work read 1.0-1.2 (part of the instruction mix
                   issues in fewer than four cycles), memory-bound ones 0.1-0.6
  valu_per_wave    SQ_INSTS_VALU / SQ_WAVES
  wait_share       SQ_WAIT_ANY / SQ_WAVE_CYCLES        share of the wave-cycles spent in s_waitcnt
  issue_wait_share SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   share spent waiting to issue an instruction
  lds_wait_share   SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES   share spent waiting to issue an LDS instruction
  lds_conflict     SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE  share of the LDS pipeline's active cycles lost to bank conflicts

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
        if "SQ_INSTS_VALU" in row and row.get("GRBM_GUI_ACTIVE"):
            row["valu_load"] = round(row["SQ_INSTS_VALU"] * 4 / (1024 * row["GRBM_GUI_ACTIVE"] / 8), 3)
        if "SQ_INSTS_VALU" in row and row.get("SQ_WAVES"):
            row["valu_per_wave"] = round(row["SQ_INSTS_VALU"] / row["SQ_WAVES"], 1)
        if row.get("SQ_WAVE_CYCLES"):
            for name, ctr in (("wait_share", "SQ_WAIT_ANY"), ("issue_wait_share", "SQ_WAIT_INST_ANY"), ("lds_wait_share", "SQ_WAIT_INST_LDS")):
                if ctr in row:
                    row[name] = round(row[ctr] / row["SQ_WAVE_CYCLES"], 3)
        if row.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in row:
            row["lds_conflict"] = round(row["SQ_LDS_BANK_CONFLICT"] / row["SQ_LDS_IDX_ACTIVE"], 3)
        doc[k] = row
    json.dump(doc, open(out, "w"), indent=1)
    for k, row in doc.items():
        print(k[:70], {a: b for a, b in row.items() if a in ("hbm_bytes", "valu_load", "valu_per_wave", "wait_share", "issue_wait_share", "lds_wait_share", "lds_conflict")})


if __name__ == "__main__":
    main()
