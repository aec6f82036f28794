# every launch of ONE steady-state grounding step (bench.py's grounding leg: 8 videos), in order:   bash benchmarks/grd_seq.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/gs
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gs -o bench -- python3 $R/bench.py --videos 8 --cpu-videos 0 --exact-steps 0 --fast-steps 0 --recall-steps 0 --legs grounding --steps 2 --warmup 1 --no-prof > /tmp/gs.log 2>&1
python3 - <<PY
import csv, glob, re
f = glob.glob("/tmp/gs/**/bench_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "postprocess" in r["Kernel_Name"]]
seg = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"]); tot = 0
for r in seg:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n).replace("void ", "")
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); tot += d
    print("%9.1f us  +%8.1f  grid %-8s wg %-4s %s" % (d / 1e3, (int(r["Start_Timestamp"]) - t0) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), n[:80]))
print("kernels of the step: %.3f ms in %d launches; wall %.3f ms" % (tot / 1e6, len(seg), (int(seg[-1]["End_Timestamp"]) - t0) / 1e6))
PY
