# per-kernel time of the training step (rocprofv3 --kernel-trace --stats):   bash benchmarks/train_probe.sh [videos] [ENV=VALUE ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-64}
shift
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/tp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp -o bench -- python3 $R/bench.py --mode train --train-videos $V --steps 5 --warmup 2 --no-prof > /tmp/tp.log 2>&1
tail -1 /tmp/tp.log | cut -c1-200
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/tp/**/bench_kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows if "copyBuffer" not in r["Name"] and "CatArray" not in r["Name"])
print("kernel ms per step (7 steps; setup copies excluded):", tot/1e6/7)
for r in rows[:32]:
    print("%-100s %7.1f/step %9.1f us avg %8.3f ms/step"%(r["Name"][:100], int(r["Calls"])/7, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6/7))
PY
