export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof && (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-timing > /tmp/prof.log 2>&1)
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows[:22]:
    per=float(r['TotalDurationNs'])/23/1e3
    print(f"{per:8.2f} us/step  calls/step {int(r['Calls'])/23:5.2f}  avg {float(r['AverageNs'])/1e3:7.2f}  {r['Name'][:64]}")
print("sum", sum(float(r['TotalDurationNs']) for r in rows)/23/1e3)
PY
