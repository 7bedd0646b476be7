#!/bin/bash
# One GPU-box visit of round 4 (edited per visit; the generic pieces are tools/gpu_round.sh, profile_round.sh, ab_*.sh).
TAG=${1:-r04x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -3
VARIANTS="_ SLS_NO_COARSE_BIN=1" REPS=2 KERNELS=bin bash tools/ab_env.sh > gpurun_out/${TAG}_ab.txt 2>&1; cat gpurun_out/${TAG}_ab.txt
