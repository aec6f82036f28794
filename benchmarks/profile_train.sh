cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_train
rm -rf $O; mkdir -p $O
cd $R
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o train -- python3 bench.py --mode train --no-prof --train-videos ${TV:-8} > $O/stdout.log 2>&1
python3 - <<'PY'
import csv,os
p=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/prof_train/stats/train_kernel_stats.csv"
rows=list(csv.DictReader(open(p)))
for r in rows[:28]:
    print("%-90s %6s %10.3f ms %9.1f us %5s%%" % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
