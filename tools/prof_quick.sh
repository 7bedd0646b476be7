#!/bin/bash
# rocprofv3 kernel-time breakdown of the default bench run (per iteration)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
R=$PWD
rm -rf /tmp/prof && (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-timing > /tmp/prof.log 2>&1)
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    per=float(r['TotalDurationNs'])/23/1e3
    print(f"{per:8.2f} us/step  calls/step {int(r['Calls'])/23:5.2f}  avg {float(r['AverageNs'])/1e3:7.2f}  {r['Name'][:64]}")
print("sum", sum(float(r['TotalDurationNs']) for r in rows)/23/1e3)
PY
