# every launch of ONE steady-state 256-video eval step, in order, with its duration (rocprofv3 kernel trace):
#   bash benchmarks/step_seq.sh [extra bench.py arguments]        (on the GPU box; output on stdout)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/seq -o bench -- python3 $R/bench.py --cpu-videos 0 --exact-steps 0 --fast-steps 0 --legs none --recall-steps 0 --no-prof --steps 4 --warmup 2 "$@" > /tmp/seq.log 2>&1
tail -1 /tmp/seq.log | cut -c1-160
python3 - <<PY
import csv, re
rows = list(csv.DictReader(open("/tmp/seq/bench_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "bbox_feat_kernel" in r["Kernel_Name"]]
seg = rows[idx[-2]:idx[-1]]
t0 = int(seg[0]["Start_Timestamp"])
tot = 0
for r in seg:
    n = r["Kernel_Name"]
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot += d
    print("%9.1f us  +%8.1f  grid %-6s %s" % (d / 1e3, (int(r["Start_Timestamp"]) - t0) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), n[:70]))
print("kernels of the step: %.3f ms in %d launches; wall %.3f ms" % (tot / 1e6, len(seg), (int(seg[-1]["End_Timestamp"]) - t0) / 1e6))
PY
