#!/bin/bash
# One profiling visit to the GPU box for a round's artefacts:   gpurun -- 'bash tools/profile_round.sh r03x'
#   <tag>_pmc_traffic.json           FETCH_SIZE / WRITE_SIZE per kernel (two --pmc passes, tools/pmc_traffic.sh)
#   <tag>_pmc_sq.{txt,json}          SQ counters (four --pmc passes, tools/pmc_sq.sh)
#   <tag>_bench.json / .err          the driver's command (python bench.py) — run AFTER the counter passes, so that the
#                                    counters it replays are the ones just measured on this build ("stale": false)
#   <tag>_bench_kernel_stats.csv     rocprofv3 --kernel-trace --stats of the same steps
# Everything lands in gpurun_out/ (merged back); copy what should be judged into profiles/.
TAG=${1:-run}
set -x
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
timeout 300 bash tools/pmc_traffic.sh && cp gpurun_out/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
timeout 300 bash tools/pmc_sq.sh > /dev/null 2>&1; cp gpurun_out/pmc_sq.txt gpurun_out/${TAG}_pmc_sq.txt; cp gpurun_out/pmc_sq.json gpurun_out/${TAG}_pmc_sq.json
cp gpurun_out/${TAG}_pmc_traffic.json gpurun_out/${TAG}_pmc_sq.json gpurun_out/${TAG}_pmc_sq.txt profiles/    # (on the box: what bench.py replays)
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.err
rm -rf /tmp/prof && (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-timing > /tmp/prof.log 2>&1)
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_bench_kernel_stats.csv \;
head -16 gpurun_out/${TAG}_bench_kernel_stats.csv
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench.json"))
print(d["value"], d["ms_per_step"], d["config"]["ms_per_iteration"], {k: v["avg_us"] for k, v in d["kernels"].items()})
print({k: v for k, v in d["roofline"].items() if k not in ("note", "valu", "valu_calibration")})
PY
