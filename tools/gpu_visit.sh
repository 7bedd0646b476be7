#!/bin/bash
# One GPU-box visit of round 6 (edited per visit; the generic pieces are tools/gpu_round.sh, profile_round.sh, ab_*.sh).
TAG=${1:-r06x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log; tail -3 gpurun_out/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 300 gpurun_out/${TAG}_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stale"], d["extras"]["dp_world1"]["allreduce_ms_per_iteration"], d["extras"]["update_model"]["wall_over_iterations"], d["cpu_baseline"]["value"])
PY
