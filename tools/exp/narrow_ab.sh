#!/bin/bash
# A/B of library variants on the narrow widths of the three decoders: tools/exp/narrow_ab.sh NAME...
set -o pipefail
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset SIMDCOMP_HIP_LIB; else export SIMDCOMP_HIP_LIB=/root/repo/simdcomp_amd/variants/lib$v.so; fi
  timeout -k 10 200 python3 /root/repo/tools/microbench.py --log2n 30 --what ${NARROW_WHAT:-unpackd1,unpack,unpackFOR} --bits ${NARROW_BITS:-1,2,3} --reps 7 > /root/repo/gpurun_out/nab_$v.log 2>&1 || { tail -5 /root/repo/gpurun_out/nab_$v.log; exit 1; }
  echo "== $v"
  grep -h "^unpack\|^pack" /root/repo/gpurun_out/nab_$v.log | python3 -c "
import sys, json
for line in sys.stdin:
    tag, body = line.split(' ', 1)
    print(tag, {k: v['frac'] for k, v in json.loads(body).items()})"
done
