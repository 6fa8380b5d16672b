#!/bin/bash
# Builds librtx from the sources of <commit> (or of the working tree: "wt") into build/exp/librtx_<name>.so [extra hipcc flags]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$1; NAME=$2; shift 2
TMP=$(mktemp -d)
if [ "$C" = "wt" ]; then mkdir -p $TMP/raytracing_amd; cp -r $ROOT/raytracing_amd/csrc $TMP/raytracing_amd/csrc; cp -r $ROOT/include $TMP/include
else git -C $ROOT archive $C raytracing_amd/csrc include | tar -x -C $TMP; fi
cd $TMP/raytracing_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form "$@" -shared -o $ROOT/build/exp/librtx_$NAME.so trace_kernel.hip rtx_api.hip rtx_host.cpp -ldl
rm -rf $TMP
echo built build/exp/librtx_$NAME.so
