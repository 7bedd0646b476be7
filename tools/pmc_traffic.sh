#!/bin/bash
# HBM traffic of every kernel of one mapping iteration from the TCC PMC counters
# (MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE need separate passes; both are in KiB;
# on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream -> calibrated below on adam_kernel,
# whose traffic is known exactly: 16 B read + 12 B written per element).
# Usage (on the GPU box):  bash tools/pmc_traffic.sh  -> gpurun_out/pmc_traffic.json
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python tools/one_iter.py 4 > /tmp/pmc_$c.log 2>&1
done
python - <<'PY'
import csv, collections, json
N = 500000
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f"/tmp/pmc_{c}/p_counter_collection.csv")):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sls::", "")
        agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in agg:
        out.setdefault(k, {})[c + "_KiB_per_launch"] = agg[k] / cnt[k]
        out[k]["launches"] = cnt[k]
adam = out["adam_kernel"]
read_true, write_true = N * 10 * 16, N * 10 * 12
f_corr = read_true / (adam["FETCH_SIZE_KiB_per_launch"] * 1024)
w_corr = write_true / (adam["WRITE_SIZE_KiB_per_launch"] * 1024)
import sys
sys.path.insert(0, ".")
from bench_support import kernel_source_hash
res = {"kernel_source_hash": kernel_source_hash(),
       "calibration": {"kernel": "adam_kernel", "true_read_bytes": read_true, "true_write_bytes": write_true,
                       "fetch_factor": f_corr, "write_factor": w_corr,
                       "note": "factor = known bytes / (counter KiB * 1024); the guide's gfx950 FETCH_SIZE correction is x2"},
       "kernels": {}}
for k, v in out.items():
    f = v.get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024
    w = v.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024
    res["kernels"][k] = {"fetch_bytes_raw": f, "write_bytes_raw": w,
                         "hbm_bytes_corrected": f * 2.0 + w, "hbm_bytes_calibrated": f * f_corr + w * w_corr}
json.dump(res, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res["calibration"]))
for k in sorted(res["kernels"]):
    if k.startswith(("render_", "adam_", "preprocess_")):
        print(k, {a: round(b / 1e6, 2) for a, b in res["kernels"][k].items()}, "MB")
PY
