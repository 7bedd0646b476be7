#!/bin/bash
# One GPU-box visit of round 4 (edited per visit; the generic pieces are tools/gpu_round.sh, profile_round.sh, ab_*.sh).
TAG=${1:-r04x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
VARIANTS="_ SLS_ORDER_AGE_ROUND2=4+SLS_ORDER_AGE_ROUND3=12+SLS_ORDER_AGE_EXTRA=48" REPS=2 KERNELS=resort bash tools/ab_env.sh > gpurun_out/${TAG}_ab.txt 2>&1; cut -c1-600 gpurun_out/${TAG}_ab.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4
