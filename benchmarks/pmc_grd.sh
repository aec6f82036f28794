# SQ counters of the kernels of the grounding leg (name filter $1, default mha_core_hd16_lds_kernel<8>):   bash benchmarks/pmc_grd.sh [filter]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F=${1:-"mha_core_hd16_lds_kernel<8>"}
O=/tmp/pmc_grd
rm -rf $O; mkdir -p $O
cd $R
P1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS"
P4="SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p$i -o pmc -- python3 bench.py --videos 8 --cpu-videos 0 --exact-steps 0 --fast-steps 0 --recall-steps 0 --legs grounding --steps 2 --warmup 1 --no-prof > $O/p$i.log 2>&1
done
F="$F" python3 - <<'PY'
import csv, glob, collections, os
flt = os.environ["F"]
tot = collections.defaultdict(float); n = collections.Counter(); big = collections.defaultdict(float)
for f in glob.glob("/tmp/pmc_grd/p*/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if flt in r["Kernel_Name"]]
    # the largest launch of that kernel (by grid) only
    g = max(int(r["Grid_Size"]) for r in rows) if rows else 0
    for r in rows:
        if int(r["Grid_Size"]) == g:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:32s} {tot[k] / max(n[k], 1):16.0f}  (avg over {n[k]} dispatch rows)")
PY
