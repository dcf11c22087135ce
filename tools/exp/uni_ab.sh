#!/bin/bash
# A/B of uniform-width kernel variants: tools/exp/uni_ab.sh "WHAT" "BITS" NAME...  (libraries built by tools/variant.sh)
set -e
mkdir -p gpurun_out/uni
WHAT=$1; BITS=$2; shift; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset SIMDCOMP_HIP_LIB; else export SIMDCOMP_HIP_LIB=simdcomp_amd/variants/lib$v.so; fi
  echo "== $v"
  timeout -k 10 300 python3 tools/microbench.py --log2n 30 --what $WHAT --bits $BITS --reps 7 2>&1 | grep -v amdgpu.ids > gpurun_out/uni/$v.log
  python3 - gpurun_out/uni/$v.log <<'PY'
import sys, json
for line in open(sys.argv[1]):
    kind, _, rest = line.partition(" ")
    if rest.startswith("{"):
        d = json.loads(rest)
        rows = [(int(b), v["frac"]) for b, v in d.items() if isinstance(v, dict) and "frac" in v and b.isdigit()]
        if rows:
            print(kind, " ".join(f"{b}:{f:.3f}" for b, f in rows), "mean %.4f min %.3f" % (sum(f for _, f in rows) / len(rows), min(f for _, f in rows)))
PY
done
