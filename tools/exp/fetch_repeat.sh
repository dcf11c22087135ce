#!/bin/bash
# FETCH_SIZE and duration of the decode launches in N fresh processes: does the slow state of a width fetch more?
cd /tmp && export TMPDIR=/tmp
N=${1:-5}
mkdir -p /root/repo/gpurun_out/fr
for i in $(seq 1 $N); do
  rm -rf /root/repo/gpurun_out/fr/$i
  timeout -s KILL 200 rocprofv3 --pmc ${2:-FETCH_SIZE} --output-format csv -d /root/repo/gpurun_out/fr/$i -- python3 /root/repo/bench.py --steps 2 --warmup 0 --no-cpu --no-h2h --no-encode --no-callers --no-traffic > /root/repo/gpurun_out/fr/$i.log 2>&1 || { tail -3 /root/repo/gpurun_out/fr/$i.log; exit 1; }
  python3 - <<PY
import csv, glob, re, collections
f = glob.glob('/root/repo/gpurun_out/fr/$i/vm/*counter_collection.csv')[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    m = re.search(r'k_unpack_fast<(\d+), 1,', r['Kernel_Name'])
    if m:
        d[int(m.group(1))].append(((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000, float(r['Counter_Value']) / 1024))
print($i, ' '.join('%d:%.0fus/%.0fMB' % (b, min(x[0] for x in d[b]), sum(x[1] for x in d[b]) / len(d[b])) for b in (8, 9, 10, 11, 12, 13, 14, 20, 26)))
PY
done
