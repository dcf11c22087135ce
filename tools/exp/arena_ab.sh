#!/bin/bash
# The bench twice per allocation scheme in fresh processes on one box: separate allocations / one arena.
cd /root/repo
for i in 1 2; do
  for a in ${ARENA_MODES:-0 1}; do
    SIMDCOMP_BENCH_ARENA=$a timeout -k 10 300 python bench.py --no-cpu --no-h2h --no-callers --no-traffic --no-encode > gpurun_out/arena_${a}_$i.json 2> gpurun_out/arena_${a}_$i.err || exit 1
    python3 - <<P
import json
l = json.load(open("gpurun_out/arena_${a}_$i.json")); pw = l["roofline"]["per_width"]
print("arena=$a run $i  agg %.1f  " % (l["roofline"]["frac"] * 100) + " ".join("%s:%.0f" % (k, v["frac"] * 100) for k, v in pw.items()))
P
  done
done
