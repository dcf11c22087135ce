#!/bin/bash
# The bench in fresh processes on one box under different settings of one environment variable:
#   tools/exp/env_ab.sh NAME VALUE...      ("-" = unset)
cd /root/repo
name=$1; shift
for i in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $name; else export $name=$v; fi
    timeout -k 10 300 python bench.py --no-cpu --no-h2h --no-callers --no-traffic --no-encode > gpurun_out/env_${v}_$i.json 2> gpurun_out/env_${v}_$i.err || exit 1
    python3 - <<P
import json
l = json.load(open("gpurun_out/env_${v}_$i.json")); pw = l["roofline"]["per_width"]
print("$name=$v run $i  agg %.1f  " % (l["roofline"]["frac"] * 100) + " ".join("%s:%.0f" % (k, v["frac"] * 100) for k, v in pw.items()))
P
  done
done
