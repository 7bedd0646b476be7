#!/bin/bash
# One GPU-box visit of round 4 (edited per visit; the generic pieces are tools/gpu_round.sh, profile_round.sh, ab_*.sh).
TAG=${1:-r04x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_timed_path.py -m gpu -q -x -k "loss_stage or checker_chain or tall" 2>&1 | tail -15
VARIANTS="_ SLS_NO_FUSED_B=1" REPS=2 KERNELS=e bash tools/ab_env.sh > gpurun_out/${TAG}_ab.txt 2>&1; cut -c1-600 gpurun_out/${TAG}_ab.txt
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -5
