cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in 0 1; do
  BIGC_PRESPLIT=$e timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp$e -o bench -- python3 $R/bench.py --cpu-videos 0 --exact-steps 0 --fast-steps 0 --recall-steps 0 --legs none --steps 5 --warmup 2 --no-prof > /tmp/pp$e.log 2>&1
  tail -1 /tmp/pp$e.log | cut -c1-120
  echo "== BIGC_PRESPLIT=$e"
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/pp$e/**/bench_kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows if "copyBuffer" not in r["Name"] and "at::native" not in r["Name"])
n=[int(r["Calls"]) for r in rows if "construct_triplets" in r["Name"]][0]          # forward passes in the trace (warm-up + timed + parity run)
print("library kernels: %.2f ms per forward pass (%d passes)" % (tot/1e6/n, n))
for r in rows[:26]:
    print("%-100s %6.1f per pass %10.1f us avg %8.2f ms/pass"%(r["Name"][:100], int(r["Calls"])/n, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6/n))
PY
done
