#!/bin/bash
# the decode leg of bench.py in fresh processes on one box, alternating between library builds:
# tools/exp/bench_alt.sh ROUNDS NAME...   ("base" = the in-tree library, others from simdcomp_amd/variants/)
cd /root/repo
R=$1; shift
mkdir -p gpurun_out/alt
for i in $(seq 1 $R); do
  for v in "$@"; do
    if [ "$v" = base ]; then unset SIMDCOMP_HIP_LIB; else export SIMDCOMP_HIP_LIB=/root/repo/simdcomp_amd/variants/lib$v.so; fi
    timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-h2h --no-encode --no-callers --no-traffic --no-for --no-mixed > gpurun_out/alt/$v.$i.json 2> gpurun_out/alt/$v.$i.err || { tail -5 gpurun_out/alt/$v.$i.err; exit 1; }
    python3 - <<PY
import json
d = json.load(open('gpurun_out/alt/$v.$i.json'))
pw = d['roofline']['per_width']
print('$v', $i, d['roofline']['frac'], ' '.join('%s:%.3f' % (k, pw[k]['frac']) for k in ('1', '2', '4', '7', '8', '9', '10', '12', '14', '16', '20', '23', '26', '32')))
PY
  done
done
