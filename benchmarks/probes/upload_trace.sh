# kernel trace of benchmarks/probes/upload_probe2.py: durations of copy_mapped_kernel and the gaps between launches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/utr -o t -- python3 $R/benchmarks/probes/upload_probe2.py ${1:-none} > /tmp/utr.log 2>&1
tail -8 /tmp/utr.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("/tmp/utr/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cp = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "copy_mapped" in r["Kernel_Name"]]
cp.sort()
print("copy_mapped launches %d: median %.1f us, p99 %.1f us, max %.1f us" % (len(cp), cp[len(cp) // 2], cp[int(len(cp) * 0.99)], cp[-1]))
gaps = []
for a, b in zip(rows[:-1], rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g > 500:
        gaps.append((round(g / 1e3, 2), a["Kernel_Name"][:30], b["Kernel_Name"][:30]))
print("gaps > 0.5 ms between consecutive kernels (ms, before, after):", gaps[:40])
print("span %.1f ms, kernels %d" % ((int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6, len(rows)))
PY
