# kernel counts and times of ONE steady-state grounding training step (bench.py's grd_train_8 leg: 8 videos), by kernel name:
#   bash benchmarks/grd_train_seq.sh        (on the GPU box; output on stdout)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gts -o bench -- python3 bench.py --gpus 1 --steps 2 --warmup 1 --cpu-videos 0 --exact-steps 0 --fast-steps 0 --recall-steps 0 --legs grounding,train --no-prof > /tmp/gts.log 2>&1
python3 - <<PY
import csv, re, collections
rows = list(csv.DictReader(open("/tmp/gts/bench_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step = from one fused-Adam launch to the next (the leg's optimizer), the last complete one
idx = [i for i, r in enumerate(rows) if "multi_tensor_apply_kernel" in r["Kernel_Name"] and "FusedOptimizer" in r["Kernel_Name"]]
# the optimizer is several launches in a row: keep the first of each run
firsts = [i for j, i in enumerate(idx) if j == 0 or idx[j - 1] != i - 1]
seg = rows[firsts[-2]:firsts[-1]]
tot = collections.Counter(); cnt = collections.Counter()
for r in seg:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n).replace("void ", "")
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot[n] += d; cnt[n] += 1
print("launches per step: %d, kernel ms: %.3f, first start to last end: %.3f ms" % (len(seg), sum(tot.values()) / 1e6, (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6))
for k, t in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print("%5d launches %9.1f us  avg %7.1f  %s" % (t, tot[k] / 1e3, tot[k] / 1e3 / t, k[:100]))
PY
