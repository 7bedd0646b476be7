#!/bin/bash
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cp splat_loam_amd/libsls_hip.so /tmp/keep.so
REPS=3 KERNELS=render bash tools/ab_bench.sh new2 new3
cp /tmp/keep.so splat_loam_amd/libsls_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_timed_path.py -m gpu -q -x > gpurun_out/r03k_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03k_pytest.log
tail -3 gpurun_out/r03k_pytest.log
