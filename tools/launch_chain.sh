#!/bin/bash
# The cost of a dependent kernel launch under different runtime settings (tools/micro/launch_chain.hip).
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
B=gpurun_tmp_launch_chain
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $B tools/micro/launch_chain.hip || exit 1
run() { echo "== $*"; env "$@" ./$B 300; }
run X=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1
run ROC_SKIP_KERNEL_ARG_COPY=1
run GPU_MAX_HW_QUEUES=1
run ROC_ACTIVE_WAIT_TIMEOUT=0
run AMD_DIRECT_DISPATCH=0
run HIP_FORCE_DEV_KERNARG=1 AMD_OPT_FLUSH=1 ROC_SKIP_KERNEL_ARG_COPY=1
