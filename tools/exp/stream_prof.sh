#!/bin/bash
# stream tests, then the byte-stream rows of the microbenchmark under rocprofv3 --kernel-trace --stats; $1 = tag
set -o pipefail
tag=$1
cd /root/repo
timeout -k 10 300 python -u -m pytest tests/test_gpu_stream.py -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/${tag}_test.log 2>&1 || { tail -15 gpurun_out/${tag}_test.log; exit 1; }
tail -1 gpurun_out/${tag}_test.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${tag}_prof -- python3 /root/repo/tools/microbench.py --log2n 30 --what stream > /root/repo/gpurun_out/${tag}_mb.log 2>&1 || { tail -5 /root/repo/gpurun_out/${tag}_mb.log; exit 1; }
grep -h "stream_" /root/repo/gpurun_out/${tag}_mb.log | cut -c1-900
python3 /root/repo/tools/kstats.py /root/repo/gpurun_out/${tag}_prof | grep -v "k_pack_fast\|k_reduce\|k_stream_copy<true>\|k_scan" | head -16
