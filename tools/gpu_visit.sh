#!/bin/bash
# One GPU-box visit of round 4 (edited per visit; the generic pieces are tools/gpu_round.sh, ab_bench.sh, ab_env.sh).
TAG=${1:-r04x}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -3 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -3 gpurun_out/${TAG}_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench.json"))
print(d["value"], d["config"]["ms_per_iteration"], d["config"]["repeated_iterations"])
e = d["extras"]
print(e["deterministic"]); print(e["single_keyframe"], e["real_sizes"], e["repeated_iterations_0_800"], e["ms_per_iteration_400_800"])
for k, v in e["dropin"].items():
    if isinstance(v, dict):
        for kk, vv in v.items():
            print(k, kk, {a: b for a, b in vv.items() if a != "kernels_us"})
PY
