#!/bin/bash
# The evidence runs of a round: GPU tests, smoke, bench line, kernel stats of the bench and of the byte-stream rows,
# FETCH_SIZE / WRITE_SIZE passes.  Everything lands under gpurun_out/final/.
set -o pipefail
cd /root/repo
O=/root/repo/gpurun_out/final
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -1 $O/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cut -c1-400 $O/bench.json
timeout -k 10 300 python tools/microbench.py --log2n 30 --what stream,mixed,query > $O/microbench.log 2>&1 || { tail -20 $O/microbench.log; exit 1; }
timeout -k 10 400 python tools/microbench.py --log2n 30 --what unpackd1,unpack,unpackFOR,packd1,pack,packFOR,maxbitsd1 > $O/entrypoints.log 2>&1 || { tail -20 $O/entrypoints.log; exit 1; }
timeout -k 10 300 python tools/microbench.py --log2n 30 --what unpack8,unpack16,pack8,pack16 --bits 1,4,8,12,16,20,24,28,32 > $O/lanes.log 2>&1 || { tail -20 $O/lanes.log; exit 1; }
timeout -k 10 300 python tools/exp/lengths.py > $O/lengths.log 2>&1 || { tail -20 $O/lengths.log; exit 1; }
echo microbench ok
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 /root/repo/bench.py --no-cpu --no-h2h --no-callers --no-traffic > $O/bench_rocprof.json 2> $O/bench_rocprof.err || { tail -5 $O/bench_rocprof.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stream_stats -- python3 /root/repo/tools/microbench.py --log2n 30 --what stream > $O/stream_rocprof.log 2>&1 || { tail -5 $O/stream_rocprof.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu --no-h2h --no-callers --no-traffic --no-for --no-mixed > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu --no-h2h --no-callers --no-traffic --no-for --no-mixed > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
cd /root/repo && PMC_WHAT=stream PMC_LOG2N=30 bash tools/pmc_passes.sh $O/pmc_stream || exit 1
echo done
