#!/bin/bash
# One GPU-box visit of round 5 (edited per visit; the generic pieces are tools/gpu_round.sh, profile_round.sh, ab_*.sh).
TAG=${1:-r05x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log; tail -4 gpurun_out/${TAG}_pytest.log
timeout 400 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 300 gpurun_out/${TAG}_bench.err
# per-launch durations of the bench's kernels (which launches of bin_direct are the slow ones)
rm -rf /tmp/prof && (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-timing > /tmp/prof.log 2>&1)
find /tmp/prof -name "*kernel_trace.csv" -exec cp {} gpurun_out/${TAG}_kernel_trace.csv \;
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_bench_kernel_stats.csv \;
head -12 gpurun_out/${TAG}_bench_kernel_stats.csv
ls -la gpurun_out/${TAG}_*
