#!/bin/bash
# SQ counters of the tile kernels (one rocprofv3 --pmc pass per counter group, --kernel-trace only).
# Usage (on the GPU box): bash tools/pmc_sq.sh  -> gpurun_out/pmc_sq.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/pmc_sq.txt
g=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE"; do
  g=$((g+1))
  rm -rf /tmp/pmcsq_$g
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmcsq_$g -o p -- python tools/one_iter.py 4 > /tmp/pmcsq_$g.log 2>&1
  python - "$g" <<'PY' >> gpurun_out/pmc_sq.txt
import csv, collections, sys, glob
g = sys.argv[1]
f = glob.glob(f"/tmp/pmcsq_{g}/**/*counter_collection.csv", recursive=True)
if not f:
    print("group", g, "no output"); sys.exit(0)
agg, cnt = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sls::", "")
    if not k.startswith(("render_", "resort_", "sort_", "emit_", "preprocess_", "consumer_", "bin_", "gather_")):
        continue
    agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for (k, c) in sorted(agg):
    print(f"{k:45s} {c:24s} {agg[(k, c)] / cnt[(k, c)]:16.0f}  per launch ({cnt[(k, c)]} launches)")
PY
done
cat gpurun_out/pmc_sq.txt
python - <<'PY'
import re, json
rows = {}
for line in open("gpurun_out/pmc_sq.txt"):
    m = re.match(r"(\S+(?:<[^>]*>)?)\s+(SQ_\w+)\s+(\d+)", line.strip())
    if m:
        rows.setdefault(m.group(1).split("<")[0], {})[m.group(2)] = int(m.group(3))
import sys
sys.path.insert(0, ".")
from bench_support import kernel_source_hash
out = {"kernel_source_hash": kernel_source_hash(),
       "note": "SQ counters per launch, tools/one_iter.py (first 4 iterations of the bench scene); SQ_*_CYCLES / ACTIVE / WAIT "
               "count SIMD issue slots (4 clocks), SQ_BUSY_CYCLES is summed over the 32 shader engines", "kernels": {}}
for k, v in rows.items():
    slots = v["SQ_BUSY_CYCLES"] / 32 * 1024 / 4      # issue slots of all SIMDs during the kernel
    out["kernels"][k] = {**v, "valu_issue_busy": round(v["SQ_ACTIVE_INST_VALU"] / slots, 3),
                         "valu_slots_per_inst": round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_INSTS_VALU"], 3),
                         "avg_waves_per_simd": round(v["SQ_WAVE_CYCLES"] / slots, 2)}
json.dump(out, open("gpurun_out/pmc_sq.json", "w"), indent=1)
PY
