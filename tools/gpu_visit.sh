#!/bin/bash
# One GPU-box visit of round 5 (edited per visit; the generic pieces are tools/gpu_round.sh, profile_round.sh, ab_*.sh).
TAG=${1:-r05x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
# per-launch durations of the bench's kernels (which launches of bin_direct are the slow ones: tools/bin_tail.py)
rm -rf /tmp/prof && (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-timing > /tmp/prof.log 2>&1)
find /tmp/prof -name "*kernel_trace.csv" -exec cp {} gpurun_out/${TAG}_kernel_trace.csv \;
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_bench_kernel_stats.csv \;
python tools/bin_tail.py gpurun_out/${TAG}_kernel_trace.csv 2>/dev/null | head -12
timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(d['value'], d['config']['ms_per_iteration'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
