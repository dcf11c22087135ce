#!/bin/bash
# A/B of library variants on the 8- / 16-lane rows: tools/exp/lanes_ab.sh NAME...   ("base" = the regular build)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset SIMDCOMP_HIP_LIB; else export SIMDCOMP_HIP_LIB=/root/repo/simdcomp_amd/variants/lib$v.so; fi
  timeout -k 10 200 python3 /root/repo/tools/microbench.py --log2n 30 --what ${LANES_WHAT:-pack8,pack16} --bits 1,4,8,16,24,32 > /root/repo/gpurun_out/lab_$v.log 2>&1 || { tail -5 /root/repo/gpurun_out/lab_$v.log; exit 1; }
  echo "== $v"
  grep -h "^pack\|^unpack" /root/repo/gpurun_out/lab_$v.log | python3 -c "
import sys, json
for line in sys.stdin:
    tag, body = line.split(' ', 1)
    print(tag, {k: v['frac'] for k, v in json.loads(body).items()})"
done
