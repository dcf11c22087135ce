#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_queries.py tests/test_gpu_fuzz.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest_query.log 2>&1
rc=$?
tail -n 8 $O/pytest_query.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/microbench.py --log2n 30 --what query --reps 5 > $O/query_new2.log 2>&1 && tail -n 1 $O/query_new2.log
