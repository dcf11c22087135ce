#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python3 -u -m pytest tests/test_lanes.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest_lanes.log 2>&1
rc=$?
tail -n 8 $O/pytest_lanes.log
[ $rc -ne 0 ] && exit $rc
for v in ""; do
  if [ -n "$v" ]; then export SIMDCOMP_HIP_LIB=simdcomp_amd/variants/lib$v.so; fi
  echo "== variant: ${v:-current}"
  timeout -k 10 300 python3 tools/microbench.py --log2n 30 --what unpack16 --bits 1,4,8,12,16,20,24,28,32 --reps 5 2>&1 | tail -n 1 || exit 1
done
