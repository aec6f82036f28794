cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gp -o bench -- python3 $R/bench.py --videos 8 --cpu-videos 0 --exact-steps 0 --fast-steps 0 --recall-steps 0 --legs grounding --steps 2 --warmup 1 --no-prof > /tmp/gp.log 2>&1
tail -1 /tmp/gp.log | python3 -c "
import json,sys
pass"
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/gp/**/bench_kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
# the grounding steps are the tail of the trace: find the last 'postprocess' kernels
names=[r["Kernel_Name"] for r in rows]
idx=[i for i,n in enumerate(names) if "postprocess" in n]
# one grounding step = from the kernel after the previous step's last postprocess to this step's last postprocess; take the last two postprocess launches (2 size classes per step)
end=idx[-1]; start=idx[-3]+1
agg={}
tot=0
for r in rows[start:end+1]:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    k=r["Kernel_Name"][:90]
    a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=d; tot+=d
span=(int(rows[end]["End_Timestamp"])-int(rows[start]["Start_Timestamp"]))/1e3
print("one grounding step: %d launches, %.1f us of kernels, %.1f us wall" % (end-start+1, tot, span))
for k,(n,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:25]:
    print("%-92s %4d %9.1f us" % (k, n, t))
PY
