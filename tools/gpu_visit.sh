#!/bin/bash
TAG=${1:-r06x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_timed_path.py -m gpu -q -x 2>&1 | tail -4 | tee gpurun_out/${TAG}_pytest.log
rm -rf /tmp/prof && (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-timing > /tmp/prof.log 2>&1)
find /tmp/prof -name "*kernel_trace.csv" -exec cp {} /tmp/${TAG}_kernel_trace.csv \;
python tools/bin_tail.py /tmp/${TAG}_kernel_trace.csv > gpurun_out/${TAG}_bin_tail.txt 2>/dev/null; head -11 gpurun_out/${TAG}_bin_tail.txt
