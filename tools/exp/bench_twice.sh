#!/bin/bash
# bench.py twice in a row on one box (full, then without the side legs, then full again): are the per-width rates stable?
cd /root/repo
mkdir -p gpurun_out/twice
timeout -k 10 600 python bench.py > gpurun_out/twice/a.json 2> gpurun_out/twice/a.err || exit 1
timeout -k 10 600 python bench.py --no-cpu --no-h2h --no-callers --no-traffic > gpurun_out/twice/b.json 2> gpurun_out/twice/b.err || exit 1
timeout -k 10 600 python bench.py > gpurun_out/twice/c.json 2> gpurun_out/twice/c.err || exit 1
python3 - <<'PY'
import json
for n in 'abc':
    d = json.load(open(f'gpurun_out/twice/{n}.json'))
    pw = d['roofline']['per_width']
    print(n, d['roofline']['frac'], d['roofline']['min_frac'], {k: round(v['frac'], 3) for k, v in pw.items() if int(k) in (1, 8, 9, 10, 11, 12, 13, 14, 15, 26, 32)}, d['box']['card0/unique_id'])
PY
