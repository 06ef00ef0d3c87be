#!/bin/bash
# Builds a variant of the library whose matrix-core search is compiled with extra flags (measurement aid):
#   tools/build_mx_variant.sh <name> "<extra hipcc flags>"   ->   variants/libsvx_<name>.so
set -eu
cd "$(dirname "$0")/../stereovision_amd/csrc"
NAME=$1; FLAGS=${2:-}
SRC=${MX_SRC:-svx_dense_mx.hip}
mkdir -p ../../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form \
    -Rpass-analysis=kernel-resource-usage $FLAGS -c $SRC -o /tmp/svx_dense_mx_$NAME.o 2> /tmp/svx_dense_mx_$NAME.usage || { tail -20 /tmp/svx_dense_mx_$NAME.usage; exit 1; }
python3 check_no_scratch.py /tmp/svx_dense_mx_$NAME.usage dense_wta_mx_kernel || echo "WARNING: variant $NAME spills"
OBJS=$(ls *.o | grep -v '^svx_dense_mx.o$' | grep -v amdgcn)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libsvx_$NAME.so $OBJS /tmp/svx_dense_mx_$NAME.o
echo "built variants/libsvx_$NAME.so"
