#!/bin/bash
# where the tile kernels' waves spend their time, at the three sizes (trace build gpurun_tmp_trace.so)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
cp splat_loam_amd/libsls_hip.so /tmp/keep.so
cp gpurun_tmp_trace.so splat_loam_amd/libsls_hip.so
for shape in ${SHAPES:-500000,64,2048 50000,64,1024 170000,64,1024}; do
  echo "== $shape"
  SLS_TRACE_DUMP=gpurun_out/trace_${shape//,/_}.npy SLS_TRACE_SHAPE=$shape timeout 200 python tools/wave_trace.py 2>&1 | grep -v "amdgpu.ids\|Warning\|ret = \|print(" | grep "phases\|^fwd\|^bwd\|resident\|longest forward\|list-scheduling\|rounds by row"
done
cp /tmp/keep.so splat_loam_amd/libsls_hip.so
