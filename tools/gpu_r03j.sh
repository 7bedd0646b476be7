#!/bin/bash
# round 3, visit j: step-loop diet (selects / ballots / padding record): parity, then A/B against the previous build
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_timed_path.py -m gpu -q -x > gpurun_out/r03j_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03j_pytest.log
tail -4 gpurun_out/r03j_pytest.log
cp splat_loam_amd/libsls_hip.so /tmp/keep.so
REPS=3 KERNELS=render bash tools/ab_bench.sh base new skipdead
cp /tmp/keep.so splat_loam_amd/libsls_hip.so
