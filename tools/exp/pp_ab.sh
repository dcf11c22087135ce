#!/bin/bash
# A/B of per-block-width decoder variants: tools/exp/pp_ab.sh "WIDTHS" NAME...   (libraries built by tools/variant.sh)
set -e
mkdir -p gpurun_out/pp
W=$1; shift
for v in base "$@"; do
  if [ "$v" = base ]; then unset SIMDCOMP_HIP_LIB; else export SIMDCOMP_HIP_LIB=simdcomp_amd/variants/lib$v.so; fi
  echo "== $v"
  timeout -k 10 200 python3 tools/exp/mixed_uniform.py $W 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pp/$v.log
done
