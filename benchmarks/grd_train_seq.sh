cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gts -o bench -- python3 bench.py --gpus 1 --steps 2 --warmup 1 --cpu-videos 0 --exact-steps 0 --fast-steps 0 --recall-steps 0 --legs grounding,train --no-prof > /tmp/gts.log 2>&1
python3 - <<PY
import csv, re, collections
rows = list(csv.DictReader(open("/tmp/gts/bench_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the grounding training steps are the last ones: find launches of grd_loss kernels
idx = [i for i, r in enumerate(rows) if "grd_loss_grad_kernel" in r["Kernel_Name"]]
print("loss-grad launches:", len(idx))
seg = rows[idx[-2]:idx[-1]]
tot = collections.Counter(); cnt = collections.Counter()
for r in seg:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n).replace("void ", "")
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot[n] += d; cnt[n] += 1
print("launches per step:", len(seg), "kernel ms:", sum(tot.values()) / 1e6, "wall ms:", (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6)
for k, t in tot.most_common(30):
    print("%9.1f us %5d  avg %7.1f  %s" % (t / 1e3, cnt[k], t / 1e3 / cnt[k], k[:90]))
PY
