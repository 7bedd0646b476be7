#!/bin/bash
# One GPU-box visit of round 4 (edited per visit; the generic pieces are tools/gpu_round.sh, ab_bench.sh, ab_env.sh).
TAG=${1:-r04x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -3 gpurun_out/${TAG}_pytest.log
timeout 300 bash tools/mem_calib.sh > gpurun_out/${TAG}_mem_calib.txt 2>&1; cat gpurun_out/${TAG}_mem_calib.txt
VARIANTS="_ SLS_FWD_ORDER=0 SLS_ORDER_AGE_EXTRA=24" REPS=2 KERNELS=render bash tools/ab_env.sh > gpurun_out/${TAG}_ab.txt 2>&1; cat gpurun_out/${TAG}_ab.txt
