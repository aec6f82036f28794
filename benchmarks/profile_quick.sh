cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r01e
rm -rf $O; mkdir -p $O
cd $R
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --cpu-videos 0 --exact-steps 0 --fast-steps 0 > $O/stats_stdout.log 2>&1
python3 - <<'PY'
import csv,os
p=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/prof_r01e/stats/bench_kernel_stats.csv"
rows=list(csv.DictReader(open(p)))
for r in rows[:24]:
    print("%-95s %6s %10.3f ms %9.1f us %5s%%" % (r['Name'][:95], r['Calls'], float(r['TotalDurationNs'])/1e6/13, float(r['AverageNs'])/1e3, r['Percentage']))
PY
