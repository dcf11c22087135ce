#!/usr/bin/env python3
"""Print the sc:: rows of a rocprofv3 --kernel-trace --stats --output-format csv run (directory given)."""
import csv
import glob
import sys

for path in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Name"]
        if "sc::" in n:
            print(f"{n[:100]:100s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs']) / 1e3:9.1f}")
