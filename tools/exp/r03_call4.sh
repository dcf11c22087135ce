#!/bin/bash
# Query tests on the run-sharing kernel, its microbenchmark against the old kernel, host-to-host alone, state counters.
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_queries.py tests/test_gpu_fuzz.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest_query.log 2>&1
rc=$?
tail -n 8 $O/pytest_query.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/microbench.py --log2n 30 --what query --reps 5 > $O/query_new.log 2>&1 && tail -n 2 $O/query_new.log &&
SIMDCOMP_HIP_LIB=simdcomp_amd/variants/libquery_old.so timeout -k 10 300 python3 tools/microbench.py --log2n 30 --what query --reps 5 > $O/query_old.log 2>&1 && tail -n 2 $O/query_old.log &&
timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-callers --no-traffic --no-for --no-mixed --no-encode > $O/h2h_alone.json 2> $O/h2h_alone.err &&
bash tools/exp/state_counters.sh $O/state
