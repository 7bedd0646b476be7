#!/bin/bash
# One GPU-box visit of round 4 (edited per visit; the generic pieces are tools/gpu_round.sh, profile_round.sh, ab_*.sh).
TAG=${1:-r04x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
VARIANTS="_ SLS_BIN_SPLIT=2" REPS=2 KERNELS=bin_direct bash tools/ab_env.sh > gpurun_out/${TAG}_ab.txt 2>&1; cut -c1-300 gpurun_out/${TAG}_ab.txt
