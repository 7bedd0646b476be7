#!/bin/bash
# One GPU-box visit of round 4 (edited per visit; the generic pieces are tools/gpu_round.sh, profile_round.sh, ab_*.sh).
TAG=${1:-r04x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python tools/soak.py 60000 64 1024 8 4000 > gpurun_out/${TAG}_soak.txt 2>&1; tail -12 gpurun_out/${TAG}_soak.txt
timeout 300 python tools/soak.py 500000 64 2048 8 1500 >> gpurun_out/${TAG}_soak.txt 2>&1; tail -8 gpurun_out/${TAG}_soak.txt
SLS_DETERMINISTIC=2 timeout 300 python tools/soak.py 170000 64 1024 5 1000 >> gpurun_out/${TAG}_soak.txt 2>&1; tail -8 gpurun_out/${TAG}_soak.txt
