#!/bin/bash
# Round-3 first GPU call: counter list, host-link probe, two-rank rehearsal of every bench leg, one plain bench run.
O=gpurun_out/r03
mkdir -p $O
(timeout -k 10 90 rocprofv3-avail list > $O/avail.txt 2>&1; echo "avail rc $?")
echo "== duplex probe" && timeout -k 10 400 python3 tools/exp/duplex_probe.py > $O/duplex.json 2> $O/duplex.err && echo ok &&
echo "== rehearsal" && SIMDCOMP_BENCH_SHARE_GPU=1 timeout -k 10 500 python3 bench.py --gpus 2 --log2n 24 --steps 3 --warmup 1 > $O/rehearsal.json 2> $O/rehearsal.err && echo ok &&
echo "== bench" && timeout -k 10 600 python3 bench.py --steps 5 --warmup 2 > $O/bench1.json 2> $O/bench1.err && echo ok
rc=$?
tail -c 600 $O/duplex.err $O/rehearsal.err $O/bench1.err 2>/dev/null
exit $rc
