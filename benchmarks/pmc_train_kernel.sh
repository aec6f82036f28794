# SQ counters of one kernel of the training step (name substring $1, default mha_bwd_fused; videos $2, default 64):
#   bash benchmarks/pmc_train_kernel.sh [kernel substring] [videos]   (on the MI355X box; averages per dispatch on stdout)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-mha_bwd_fused}
V=${2:-64}
O=/tmp/pmc_tk
rm -rf $O; mkdir -p $O
cd $R
P1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS"
P4="SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p$i -o pmc -- python3 bench.py --mode train --train-videos $V --steps 2 --warmup 1 --no-prof > $O/p$i.log 2>&1
done
tail -1 $O/p1.log | cut -c1-160
K="$K" python3 - <<'PY'
import csv, glob, collections, os
key = os.environ["K"]
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("/tmp/pmc_tk/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"] and int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0) >= 0:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print("kernel substring:", key)
for k in sorted(tot):
    print(f"{k:32s} {tot[k] / max(n[k], 1):16.0f}  (avg over {n[k]} dispatch rows)")
PY
