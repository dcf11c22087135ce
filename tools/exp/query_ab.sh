#!/bin/bash
# A/B of query-kernel variants on the batched search / select rows: tools/exp/query_ab.sh NAME...
set -e
mkdir -p gpurun_out/q
for v in "$@"; do
  if [ "$v" = base ]; then unset SIMDCOMP_HIP_LIB; else export SIMDCOMP_HIP_LIB=simdcomp_amd/variants/lib$v.so; fi
  timeout -k 10 300 python3 tools/microbench.py --log2n 30 --what query --reps 5 2>&1 | grep -v amdgpu.ids > gpurun_out/q/$v.log
  python3 - gpurun_out/q/$v.log $v <<'PY'
import sys, json
for line in open(sys.argv[1]):
    kind, _, rest = line.partition(" ")
    if kind == "query":
        d = json.loads(rest)
        print(sys.argv[2], " ".join(f"{k}:{v['ms']}" for k, v in d.items() if isinstance(v, dict)))
PY
done
