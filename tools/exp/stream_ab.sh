#!/bin/bash
# A/B of library variants on the byte-stream rows: tools/exp/stream_ab.sh NAME...   ("base" = the regular build)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
export MB_NOCHECK=$MB_NOCHECK
for v in "$@"; do
  if [ "$v" = base ]; then unset SIMDCOMP_HIP_LIB; else export SIMDCOMP_HIP_LIB=/root/repo/simdcomp_amd/variants/lib$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ab_$v -- python3 /root/repo/tools/microbench.py --log2n 30 --what stream > /root/repo/gpurun_out/ab_$v.log 2>&1 || { tail -5 /root/repo/gpurun_out/ab_$v.log; exit 1; }
  echo "== $v"
  grep -h "^stream " /root/repo/gpurun_out/ab_$v.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().split(' ', 1)[1])
print({k: v['ms'] for k, v in d.items() if isinstance(v, dict) and k in ('stream_index', 'chain_offsets', 'unpack_stream', 'stream_decode', 'stream_decode_plain', 'stream_compress', 'stream_build', 'stream_unbuild')})"
  python3 /root/repo/tools/kstats.py /root/repo/gpurun_out/ab_$v | grep "k_index\|k_compress" | cut -c1-40,100-140
done
