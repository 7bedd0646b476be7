#!/bin/bash
TAG=${1:-r06x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest "tests/test_fused_mapper.py" -m gpu -q -x -s 2>&1 | grep -v amdgpu | tail -30 | tee gpurun_out/${TAG}_pytest.log
