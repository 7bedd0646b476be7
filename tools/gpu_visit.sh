#!/bin/bash
# One GPU-box visit of round 4 (edited per visit; the generic pieces are tools/gpu_round.sh, profile_round.sh, ab_*.sh).
TAG=${1:-r04x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cp splat_loam_amd/libsls_hip.so /tmp/keep.so
VARIANTS="hd mg" bash tools/ab_sizes.sh > gpurun_out/${TAG}_ab.txt 2>&1; cut -c1-300 gpurun_out/${TAG}_ab.txt
cp gpurun_tmp_mg.so splat_loam_amd/libsls_hip.so
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
cp /tmp/keep.so splat_loam_amd/libsls_hip.so
