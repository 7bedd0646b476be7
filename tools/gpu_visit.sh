#!/bin/bash
# One GPU-box visit of round 5 (edited per visit; the generic pieces are tools/gpu_round.sh, profile_round.sh, ab_*.sh).
TAG=${1:-r05x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile_round.log 2>&1; tail -4 gpurun_out/${TAG}_profile_round.log
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log; tail -3 gpurun_out/${TAG}_pytest.log
timeout 120 python tools/g7_probe.py > gpurun_out/${TAG}_g7_probe.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
# per-launch durations (which launches of bin_direct are the slow ones: tools/bin_tail.py)
rm -rf /tmp/prof && (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-timing > /tmp/prof.log 2>&1)
find /tmp/prof -name "*kernel_trace.csv" -exec cp {} /tmp/${TAG}_kernel_trace.csv \;
python tools/bin_tail.py /tmp/${TAG}_kernel_trace.csv > gpurun_out/${TAG}_bin_tail.txt 2>/dev/null; head -11 gpurun_out/${TAG}_bin_tail.txt
