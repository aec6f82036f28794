cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in 0 1; do
  BIGC_CHAINED=$e timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps$e -o bench -- python3 $R/bench.py --cpu-videos 0 --exact-steps 0 --fast-steps 0 --legs none --steps 5 --warmup 2 > /tmp/ps$e.log 2>&1
  tail -1 /tmp/ps$e.log | cut -c1-200
  echo "== BIGC_CHAINED=$e"
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/ps$e/bench_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows if "copyBuffer" not in r["Name"])
for r in rows[:22]:
    print("%-95s %6s %10.1f us avg %8.2f ms total"%(r["Name"][:95], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
