#!/bin/bash
TAG=${1:-r06x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for s in "500000 64 2048" "170000 64 1024" "50000 64 1024"; do timeout 120 python tools/fwd_stats.py $s; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_fwd_stats.txt
