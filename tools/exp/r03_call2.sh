#!/bin/bash
# Full GPU test-suite, then a plain bench run.
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 1000 python3 -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > $O/pytest_gpu.log 2>&1
rc=$?
tail -n 15 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 bench.py --steps 5 --warmup 2 > $O/bench2.json 2> $O/bench2.err
rc=$?
tail -c 400 $O/bench2.err
exit $rc
