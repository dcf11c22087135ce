#!/bin/bash
# The bench in fresh processes on one box with every slice of the single allocation at an odd multiple of A bytes.
cd /root/repo
for i in 1 2; do
  for a in ${ALIGNS:-0 4096 65536 1048576}; do
    SIMDCOMP_BENCH_ALIGN=$a timeout -k 10 300 python bench.py --no-cpu --no-h2h --no-callers --no-traffic --no-encode > gpurun_out/align_${a}_$i.json 2> gpurun_out/align_${a}_$i.err || exit 1
    python3 - <<P
import json
l = json.load(open("gpurun_out/align_${a}_$i.json")); pw = l["roofline"]["per_width"]
print("align=$a run $i  agg %.1f  " % (l["roofline"]["frac"] * 100) + " ".join("%s:%.0f" % (k, v["frac"] * 100) for k, v in pw.items()))
P
  done
done
