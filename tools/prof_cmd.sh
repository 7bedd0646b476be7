#!/bin/bash
# rocprofv3 kernel-time breakdown of an arbitrary command: tools/prof_cmd.sh <calls> <cmd...>
export TMPDIR=/tmp
CALLS=$1; shift
R=$PWD
rm -rf /tmp/prof && (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- "$@" > /tmp/prof.log 2>&1)
CALLS=$CALLS python - <<'PY'
import csv, glob, os
n = float(os.environ["CALLS"])
f = glob.glob('/tmp/prof/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print("%9.1f us/call  launches/call %5.1f  avg %8.2f  %s" % (float(r['TotalDurationNs']) / n / 1e3, int(r['Calls']) / n, float(r['AverageNs']) / 1e3, r['Name'][:70]))
print("sum", sum(float(r['TotalDurationNs']) for r in rows) / n / 1e3)
PY
