# Does less HBM traffic buy clock on this power-limited chip?  
# The same 256-video eval step in two forms that differ in HBM traffic and nothing else that matters to the matrix pipe:
#   default            LayerNorms as separate passes (33 tensor passes per decoder layer)
#   BIGC_LN_FUSED=1    the three LayerNorms of a decoder layer as epilogues of the GEMMs in front of them (3.4 GB less traffic per step)
# Per form: kernel durations (trace), GRBM_GUI_ACTIVE + SQ_VALU_MFMA_BUSY_CYCLES (clock and matrix-pipe duty of the K = 512 decoder GEMMs),
# FETCH_SIZE and WRITE_SIZE in separate passes (whole-step HBM bytes, FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950).
#   bash benchmarks/traffic_vs_clock.sh        (on the MI355X box; summary on stdout)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=/tmp/tvc
rm -rf $O; mkdir -p $O
cd $R
HEAD="--steps 3 --warmup 2 --no-prof --cpu-videos 0 --exact-steps 0 --fast-steps 0 --recall-steps 0 --legs none"
for form in default fused; do
  if [ $form = fused ]; then export BIGC_LN_FUSED=1; else unset BIGC_LN_FUSED; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/${form}_t -o t -- python3 bench.py $HEAD > $O/${form}_t.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/${form}_sq -o pmc -- python3 bench.py $HEAD > $O/${form}_sq.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${form}_$c -o pmc -- python3 bench.py $HEAD > $O/${form}_$c.log 2>&1
  done
done
unset BIGC_LN_FUSED
python3 - <<'PY'
import csv, glob, collections, json
MARK = "bbox_feat_kernel"
def rows(path):
    f = glob.glob(path, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
def last_step(rs, key="Dispatch_Id"):
    marks = sorted(set(int(r[key]) for r in rs if MARK in r["Kernel_Name"]))
    return marks[-3], marks[-2]
out = {}
for form in ("default", "fused"):
    tr = rows("/tmp/tvc/%s_t/**/*kernel_trace.csv" % form)
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(tr) if MARK in r["Kernel_Name"]]
    seg = tr[idx[-3]:idx[-2]]
    dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    step_ms = sum(dur(r) for r in seg) / 1e6
    dec = [r for r in seg if "gemm_ps_kernel<0" in r["Kernel_Name"]]
    ln = [r for r in seg if "layernorm" in r["Kernel_Name"]]
    sq = rows("/tmp/tvc/%s_sq/**/*counter_collection.csv" % form)
    a, b = last_step(sq)
    cyc = collections.defaultdict(float); n = collections.Counter()
    ids = set()
    for r in sq:
        d = int(r["Dispatch_Id"])
        if a <= d < b and "gemm_ps_kernel<0" in r["Kernel_Name"]:
            cyc[r["Counter_Name"]] += float(r["Counter_Value"]); ids.add(d)
    # durations of the same dispatches in the SAME (counter) run
    sqt = rows("/tmp/tvc/%s_sq/**/*kernel_trace.csv" % form)
    dsum = sum(dur(r) for r in sqt if int(r["Dispatch_Id"]) in ids)
    hb = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        pr = rows("/tmp/tvc/%s_%s/**/*counter_collection.csv" % (form, c))
        a2, b2 = last_step(pr)
        hb[c] = sum(float(r["Counter_Value"]) for r in pr if a2 <= int(r["Dispatch_Id"]) < b2 and r["Counter_Name"] == c) * 1024
    out[form] = dict(step_kernel_ms=round(step_ms, 3), launches=len(seg), decoder_gemm_ms=round(sum(dur(r) for r in dec) / 1e6, 3), decoder_gemm_launches=len(dec),
                     layernorm_ms=round(sum(dur(r) for r in ln) / 1e6, 3),
                     hbm_GB_per_step=round((2 * hb["FETCH_SIZE"] + hb["WRITE_SIZE"]) / 1e9, 2),
                     decoder_gemm_clock_GHz=round(cyc["GRBM_GUI_ACTIVE"] / 8 / max(dsum, 1), 3),
                     decoder_gemm_mfma_busy=round(cyc["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc["GRBM_GUI_ACTIVE"] / 8) if cyc["GRBM_GUI_ACTIVE"] else 0, 3))
print(json.dumps(out, indent=1))
PY
