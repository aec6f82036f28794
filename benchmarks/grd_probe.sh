# per-kernel times of the grounding leg: bash benchmarks/grd_probe.sh (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/grd -o bench -- python3 $R/bench.py --videos 32 --steps 2 --warmup 1 --cpu-videos 0 --exact-steps 0 --fast-steps 0 --no-prof --legs grounding > /tmp/grd.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/grd/bench_kernel_trace.csv")))
# the grounding leg = everything after the last query_input_kernel group: take the last 7 grd steps
names=[r["Kernel_Name"] for r in rows]
idx=[i for i,n in enumerate(names) if "query_input_kernel" in n]
seg=rows[idx[-2]:idx[-1]]
import collections
per=collections.defaultdict(lambda:[0,0.0])
t0=int(seg[0]["Start_Timestamp"]); t1=int(seg[-1]["End_Timestamp"])
for r in seg:
    d=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    per[r["Kernel_Name"][:90]][0]+=1; per[r["Kernel_Name"][:90]][1]+=d
print("one grounding step: wall %.3f ms, kernels %.3f ms, %d launches"%((t1-t0)/1e6, sum(v[1] for v in per.values())/1e6, len(seg)))
for k,(n,t) in sorted(per.items(), key=lambda kv:-kv[1][1])[:24]:
    print("%-92s %3d %8.1f us"%(k,n,t/1e3))
PY
