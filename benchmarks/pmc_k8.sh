# fc_bbox2enti.0 (K = 8, models/model_0v10.py:405-420,446) as its own launch in front of fc_bbox2enti.2 vs folded into the staging code of that GEMM
# (BIGC_PRE8_FUSED=1): durations and SQ counters of both forms on the 256-video eval step, same front-end otherwise (BIGC_CHAINED=0: the in-loop-split
# GEMM is the one that has the fused staging).      bash benchmarks/pmc_k8.sh > gpurun_out/r05_pmc_k8.log      (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=/tmp/pmc_k8
rm -rf $O; mkdir -p $O
cd $R
P1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
export BIGC_CHAINED=0
for form in 0 1; do
  export BIGC_PRE8_FUSED=$form
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/f${form}_t -o t -- python3 bench.py --steps 3 --warmup 1 --no-prof --cpu-videos 0 --exact-steps 0 --fast-steps 0 --recall-steps 0 --legs none > $O/f${form}_t.log 2>&1
  i=0
  for P in "$P1" "$P2"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/f${form}_p$i -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-prof --cpu-videos 0 --exact-steps 0 --fast-steps 0 --recall-steps 0 --legs none > $O/f${form}_p$i.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
def trace(form):
    f = glob.glob("/tmp/pmc_k8/f%d_t/**/*kernel_trace.csv" % form, recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    return rows
def pick(rows, form):
    """dispatch ids / durations of (the K = 8 kernel, the GEMM right behind it) resp. the fused GEMM, one pair per step"""
    out = []
    for i, r in enumerate(rows):
        n = r["Kernel_Name"]
        if form == 0 and "linear_k8_kernel" in n:
            out.append((r, rows[i + 1]))
        if form == 1 and "linear_split_kernel" in n and ", true, 0>" in n:
            out.append((r,))
    return out
for form in (0, 1):
    rows = trace(form)
    sel = pick(rows, form)
    print("== form %d (%s): %d steps" % (form, "K = 8 kernel + GEMM" if form == 0 else "K = 8 layer inside the GEMM's staging", len(sel)))
    names = [r["Kernel_Name"][:90] for r in sel[-1]]
    for j, nm in enumerate(names):
        d = [int(s[j]["End_Timestamp"]) - int(s[j]["Start_Timestamp"]) for s in sel[1:]]
        print("   %-90s %9.1f us (avg of %d)" % (nm, sum(d) / len(d) / 1e3, len(d)))
    want = set(names)
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob("/tmp/pmc_k8/f%d_p*/**/*counter_collection.csv" % form, recursive=True):
        rws = list(csv.DictReader(open(f)))
        # the same selection rule on the counter rows (dispatch order = Dispatch_Id)
        by_disp = collections.OrderedDict()
        for r in sorted(rws, key=lambda r: int(r["Dispatch_Id"])):
            by_disp.setdefault(int(r["Dispatch_Id"]), []).append(r)
        ids = list(by_disp)
        for k, d in enumerate(ids):
            n = by_disp[d][0]["Kernel_Name"]
            take = []
            if form == 0 and "linear_k8_kernel" in n:
                take = [d] + ([ids[k + 1]] if k + 1 < len(ids) else [])
            if form == 1 and "linear_split_kernel" in n and ", true, 0>" in n:
                take = [d]
            for t in take:
                for r in by_disp[t]:
                    tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    for k in sorted(tot):
        steps = max(1, cnt[k] // (2 if form == 0 else 1))
        print("   %-28s %16.0f per step (summed over the form's launches)" % (k, tot[k] / steps))
PY
