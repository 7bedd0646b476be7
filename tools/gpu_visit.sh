#!/bin/bash
# One GPU-box visit of round 4 (edited per visit; the generic pieces are tools/gpu_round.sh, ab_bench.sh, ab_env.sh).
TAG=${1:-r04x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -3 gpurun_out/${TAG}_pytest.log
REPS=2 bash tools/ab_bench.sh base new > gpurun_out/${TAG}_ab.txt 2>&1; cat gpurun_out/${TAG}_ab.txt
cp gpurun_tmp_new.so splat_loam_amd/libsls_hip.so
VARIANTS="_" REPS=1 SHAPES="170000,64,1024 50000,64,1024" bash tools/ab_env.sh >> gpurun_out/${TAG}_ab.txt 2>&1; tail -2 gpurun_out/${TAG}_ab.txt
R=$PWD
rm -rf /tmp/prof && (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-timing > /tmp/prof.log 2>&1)
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_bench_kernel_stats.csv \;
cut -d, -f1-4 gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-150 | head -16
timeout 400 python tools/c3_noise.py > gpurun_out/${TAG}_c3_noise.txt 2>&1; tail -40 gpurun_out/${TAG}_c3_noise.txt
