#!/bin/bash
# Build a variant of the library with extra compile-time definitions, for A/B timing on the GPU box:
#   tools/variant.sh NAME FILE "-DSC_REDUCE_U=2 ..."     (FILE: codec_kernels | stream_kernels | query_kernels)
# -> simdcomp_amd/variants/libNAME.so (the other objects come from the regular build); select it at run time with
#   SIMDCOMP_HIP_LIB=simdcomp_amd/variants/libNAME.so python tools/microbench.py ...
set -e
cd "$(dirname "$0")/../simdcomp_amd/csrc"
NAME=$1; FILE=$2; DEFS=$3
mkdir -p ../variants build
make -s build/exports.map $(for f in codec_kernels stream_kernels query_kernels device_api host_api scalar_access host_arith; do echo build/$f.o; done)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DSIMDCOMP_HIP_BUILDING $DEFS --offload-arch=gfx950 -c $FILE.hip -o ../variants/$NAME.o
OBJS=""
for f in codec_kernels stream_kernels query_kernels device_api host_api scalar_access host_arith; do
  if [ "$f" = "$FILE" ]; then OBJS="$OBJS ../variants/$NAME.o"; else OBJS="$OBJS build/$f.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,--version-script=build/exports.map -o ../variants/lib$NAME.so $OBJS
rm -f ../variants/$NAME.o
echo built simdcomp_amd/variants/lib$NAME.so
