#!/bin/bash
# One GPU-box visit of round 4 (edited per visit; the generic pieces are tools/gpu_round.sh, ab_bench.sh, ab_env.sh).
TAG=${1:-r04x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -3 gpurun_out/${TAG}_pytest.log
cp splat_loam_amd/libsls_hip.so /tmp/keep.so
cp gpurun_tmp_trace.so splat_loam_amd/libsls_hip.so
for split in 1 4; do
  for shape in "500000 64 2048" "50000 64 1024"; do
    echo "#### SLS_BIN_SPLIT=$split $shape"
    SLS_BIN_SPLIT=$split timeout 120 python tools/bin_trace.py $shape 2>&1 | grep -v "amdgpu.ids\|Warning"
  done
done > gpurun_out/${TAG}_bin_trace.txt 2>&1
cp /tmp/keep.so splat_loam_amd/libsls_hip.so
cat gpurun_out/${TAG}_bin_trace.txt
