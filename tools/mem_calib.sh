#!/bin/bash
# What FETCH_SIZE / WRITE_SIZE report for access patterns with a known number of accesses (tools/micro/mem_calib.hip):
#   hipcc --offload-arch=gfx950 -O3 tools/micro/mem_calib.hip -o gpurun_tmp_mem_calib      (here)
#   gpurun -- 'bash tools/mem_calib.sh'   ->  gpurun_out/mem_calibration.json
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
BIN=$PWD/gpurun_tmp_mem_calib
[ -x "$BIN" ] || { echo "build $BIN first"; exit 1; }
timeout 120 "$BIN" > /tmp/mem_plain.json || { cat /tmp/mem_plain.json; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/memc_$c
  (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/memc_$c -o p -- "$BIN" > /tmp/memc_$c.log 2>&1)
done
python - <<'PY'
import csv, glob, json
plain = json.load(open("/tmp/mem_plain.json"))
cases = {c["kernel"]: c for c in plain["cases"] if c["accesses"]}
out = {"note": "counter bytes = reported KiB x 1024, uncorrected; per access = / accesses.  stream_*: every byte of a 2 GiB array once; "
               "gather/scatter_lines<T>: ONE access of sizeof(T) per 128-byte line, every line once, pseudo-random order (no reuse, "
               "working set 2 GiB); scatter_all8: every 8-byte slot of a 64 MiB array once in pseudo-random order", "rows": []}
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/memc_{c}/**/*counter_collection.csv", recursive=True)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c:
            continue
        k = (r["Kernel_Name"].split("(")[0].replace("void ", "").replace("HIP_vector_type<unsigned int, 4u> ", "uint4")
             .replace("HIP_vector_type<unsigned int, 2u> ", "uint2").replace("HIP_vector_type<unsigned int, 4u>", "uint4")
             .replace("HIP_vector_type<unsigned int, 2u>", "uint2").replace("unsigned int", "uint32_t").strip())
        vals.setdefault(k, {}).setdefault(c, 0.0)
        vals[k][c] += float(r["Counter_Value"]) * 1024
for k, c in cases.items():
    v = vals.get(k, {})
    out["rows"].append({"kernel": k, "accesses": c["accesses"], "bytes_per_access": c["bytes_per_access"],
                        "useful_bytes": c["accesses"] * c["bytes_per_access"],
                        "FETCH_SIZE_bytes": v.get("FETCH_SIZE"), "WRITE_SIZE_bytes": v.get("WRITE_SIZE"),
                        "fetch_per_access": (v.get("FETCH_SIZE", 0.0) / c["accesses"]) if v.get("FETCH_SIZE") is not None else None,
                        "write_per_access": (v.get("WRITE_SIZE", 0.0) / c["accesses"]) if v.get("WRITE_SIZE") is not None else None})
json.dump(out, open("gpurun_out/mem_calibration.json", "w"), indent=1)
for r in out["rows"]:
    print(f"{r['kernel']:28s} {r['bytes_per_access']:3d} B x {r['accesses']:>10d}: FETCH {r['fetch_per_access'] or 0:8.2f} B/access  WRITE {r['write_per_access'] or 0:8.2f} B/access")
PY
