#!/bin/bash
# the decode leg of bench.py in N fresh processes on one box: how much do single widths move from process to process?
# usage: tools/exp/bench_repeat.sh N [variant library]
cd /root/repo
N=${1:-6}
[ -n "$2" ] && export SIMDCOMP_HIP_LIB=/root/repo/simdcomp_amd/variants/lib$2.so
mkdir -p gpurun_out/repeat
for i in $(seq 1 $N); do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-h2h --no-encode --no-callers --no-traffic > gpurun_out/repeat/$i.json 2> gpurun_out/repeat/$i.err || { tail -5 gpurun_out/repeat/$i.err; exit 1; }
  python3 - <<PY
import json
d = json.load(open('gpurun_out/repeat/$i.json'))
pw = d['roofline']['per_width']
print($i, d['roofline']['frac'], ' '.join('%s:%.3f' % (k, pw[k]['frac']) for k in ('1', '4', '7', '8', '9', '10', '11', '12', '13', '14', '15', '16', '20', '26', '32')))
PY
done
