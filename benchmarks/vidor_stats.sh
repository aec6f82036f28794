# per-kernel stats of bench.py's VidOR classification leg (16 videos x N = 180, exp5 dims):   bash benchmarks/vidor_stats.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/vs
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vs -o bench -- python3 $R/bench.py --videos 8 --cpu-videos 0 --exact-steps 0 --fast-steps 0 --recall-steps 0 --legs vidor --steps 2 --warmup 1 --no-prof > /tmp/vs.log 2>&1
tail -1 /tmp/vs.log | cut -c1-120
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/vs/**/bench_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:26]:
    print("%6s%% %9.1f us avg %5s calls  %s" % (r["Percentage"], float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:100]))
PY
