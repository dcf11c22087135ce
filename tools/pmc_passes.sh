#!/bin/bash
# Four counter groups, one rocprofv3 run each (gfx950 takes 8 SQ counters, or FETCH_SIZE, or WRITE_SIZE, per run).
# usage: tools/pmc_passes.sh OUTDIR   (run on the GPU box from the repository root)
set -e
OUT=$(realpath -m "$1")
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
run() {
  timeout -s KILL 170 rocprofv3 --pmc $2 --output-format csv -d "$OUT/$1" -- python3 "$REPO/tools/pmc_workload.py" > "$OUT.$1.log" 2>&1
  echo "pass $1 ok"
}
mkdir -p "$OUT"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
run sq2 "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
