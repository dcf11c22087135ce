#!/bin/bash
# One bounded experiment on the per-process "low / high" state of the decoder (widths 8..15 and 20..29 run 3-5 points
# slower in some processes): bench.py's decode steps under rocprofv3 --pmc, four counter groups x two allocation modes
# (SIMDCOMP_BENCH_ARENA=0: an allocation per array, which came up low in 8 of 8 processes; 2: one allocation, high in
# 12 of 15).  Each process prints its own per-width fractions, so every counter row carries the state it was taken in.
# usage: tools/exp/state_counters.sh OUTDIR      (GPU box, from the repository root); then tools/exp/state_summary.py
O=$(realpath -m "$1")
REPO=$(pwd)
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
P1="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_UTCL2_BUSY GRBM_EA_BUSY"
P2="TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum"
P3="TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
P4="TCC_HIT_sum TCC_MISS_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_LATENCY_FIFO_FULL_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TD_TC_STALL_sum TD_TD_BUSY_sum"
i=0
for pass in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  for mode in 0 2; do
    SIMDCOMP_BENCH_ARENA=$mode timeout -s KILL 150 rocprofv3 --pmc $pass --output-format csv -d "$O/p${i}_m${mode}" -- \
      python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu --no-h2h --no-callers --no-traffic --no-for --no-mixed --no-encode \
      > "$O/p${i}_m${mode}.json" 2> "$O/p${i}_m${mode}.err" || { echo "pass $i mode $mode failed"; tail -5 "$O/p${i}_m${mode}.err"; exit 1; }
    echo "pass $i mode $mode ok"
  done
done
# the same two modes without the profiler, for the fractions an undisturbed process shows
for mode in 0 2; do
  SIMDCOMP_BENCH_ARENA=$mode timeout -k 10 150 python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu --no-h2h --no-callers --no-traffic --no-for --no-mixed --no-encode \
    > "$O/plain_m${mode}.json" 2> "$O/plain_m${mode}.err" || exit 1
  echo "plain mode $mode ok"
done
