#!/bin/bash
# A/B of per-block-width decoder variants on the mixed and byte-stream rows: tools/exp/pp_ab2.sh NAME...
set -e
mkdir -p gpurun_out/pp
for v in base "$@"; do
  if [ "$v" = base ]; then unset SIMDCOMP_HIP_LIB; else export SIMDCOMP_HIP_LIB=simdcomp_amd/variants/lib$v.so; fi
  echo "== $v"
  timeout -k 10 200 python3 tools/microbench.py --log2n 30 --what mixed,stream --reps 7 2>&1 | grep -v amdgpu.ids > gpurun_out/pp/mb_$v.log
  python3 - gpurun_out/pp/mb_$v.log <<'PY'
import sys, json
for line in open(sys.argv[1]):
    kind, _, rest = line.partition(" ")
    if kind in ("mixed", "stream"):
        d = json.loads(rest)
        print(kind, d.get("avg_bits"), {k: v["ms"] for k, v in d.items() if isinstance(v, dict) and k in ("unpack_mixed", "unpack_stream", "stream_decode", "stream_decode_plain", "pack_mixed", "pack_stream", "stream_compress")})
PY
  if [ -z "$PP_NO_UNIFORM" ]; then timeout -k 10 200 python3 tools/exp/mixed_uniform.py 4 6 7 8 9 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pp/$v.log; fi
done
