# every launch of ONE steady-state step of the grounding leg (8 videos, 1098 queries), in order, with its duration (rocprofv3 kernel trace):
#   bash benchmarks/grd_seq.sh        (on the GPU box; output on stdout)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gseq -o bench -- python3 bench.py --videos 8 --cpu-videos 0 --exact-steps 0 --fast-steps 0 --recall-steps 0 --legs grounding --steps 2 --warmup 1 --no-prof > /tmp/gseq.log 2>&1
tail -1 /tmp/gseq.log | cut -c1-200
python3 - <<PY
import csv, re
rows = list(csv.DictReader(open("/tmp/gseq/bench_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "postprocess_kernel" in r["Kernel_Name"]]
seg = rows[idx[-2] + 1:idx[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
tot = 0
for r in seg:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot += d
    print("%9.1f us  +%8.1f  grid %-9s wg %-4s %s" % (d / 1e3, (int(r["Start_Timestamp"]) - t0) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), n[:90]))
print("kernels of the step: %.3f ms in %d launches; first start to last end %.3f ms" % (tot / 1e6, len(seg), (int(seg[-1]["End_Timestamp"]) - t0) / 1e6))
PY
