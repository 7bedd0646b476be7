set -x
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/r01k_bench.json 2> gpurun_out/r01k_bench.err
tail -c 600 gpurun_out/r01k_bench.err
rm -rf /tmp/prof && (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-timing > /tmp/prof.log 2>&1)
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r01k_bench_kernel_stats.csv \;
head -12 gpurun_out/r01k_bench_kernel_stats.csv
bash tools/pmc_traffic.sh && cp gpurun_out/pmc_traffic.json gpurun_out/r01k_pmc_traffic.json
