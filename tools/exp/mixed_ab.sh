#!/bin/bash
# A/B of library variants on the per-block-width rows: tools/exp/mixed_ab.sh NAME...   ("base" = the regular build)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset SIMDCOMP_HIP_LIB; else export SIMDCOMP_HIP_LIB=/root/repo/simdcomp_amd/variants/lib$v.so; fi
  timeout -k 10 200 python3 /root/repo/tools/microbench.py --log2n 30 --what stream,mixed --reps 9 > /root/repo/gpurun_out/mab_$v.log 2>&1 || { tail -5 /root/repo/gpurun_out/mab_$v.log; exit 1; }
  echo "== $v"
  grep -h "^stream \|^mixed " /root/repo/gpurun_out/mab_$v.log | python3 -c "
import sys, json
for line in sys.stdin:
    tag, body = line.split(' ', 1)
    d = json.loads(body)
    print(tag, {k: v['ms'] for k, v in d.items() if isinstance(v, dict) and k in ('unpack_mixed', 'pack_mixed', 'pack_stream', 'stream_compress', 'unpack_stream', 'stream_decode', 'stream_decode_plain', 'chain_offsets')})"
done
