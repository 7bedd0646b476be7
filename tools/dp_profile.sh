#!/bin/bash
# Kernel table of the one-rank exchange (extras.dp_world1): rocprofv3 --kernel-trace --stats of `python bench_extras.py dp_world1`
#   gpurun -- 'bash tools/dp_profile.sh r08b'  ->  gpurun_out/<tag>_dp_kernel_stats.csv, <tag>_dp.json
TAG=${1:-run}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
timeout 600 python bench_extras.py dp_world1 > gpurun_out/${TAG}_dp.json 2> gpurun_out/${TAG}_dp.err; tail -c 300 gpurun_out/${TAG}_dp.err; cat gpurun_out/${TAG}_dp.json
rm -rf /tmp/prof && (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/bench_extras.py dp_world1 > /tmp/prof.log 2>&1)
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_dp_kernel_stats.csv \;
cut -c1-90 gpurun_out/${TAG}_dp_kernel_stats.csv | head -5
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/${TAG}_dp_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.3: print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "%9.1f" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
PY
