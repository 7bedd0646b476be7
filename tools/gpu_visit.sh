#!/bin/bash
TAG=${1:-r06x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
SHAPES="170000,64,1024" VARIANTS="SLS_BWD_SPLIT=0 SLS_BWD_SPLIT=1+SLS_BWD_SPLIT_MIN=160 SLS_BWD_SPLIT=1+SLS_BWD_SPLIT_MIN=190 SLS_BWD_SPLIT=1+SLS_BWD_SPLIT_MIN=220 SLS_BWD_SPLIT=1+SLS_BWD_SPLIT_MIN=260" KERNELS=render REPS=2 bash tools/ab_env.sh 2>&1 | tee gpurun_out/${TAG}_ab_split.txt
