# SQ counters of one kernel of the 256-video eval step (name filter $1):   bash benchmarks/pmc_eval_kernel.sh "mha_core_lds16_kernel<12"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F=${1:-"mha_core_lds16_kernel<12"}
O=/tmp/pmc_ek
rm -rf $O; mkdir -p $O
cd $R
P1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS"
P4="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS"
i=0
for P in "$P1" "$P3" "$P4"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p$i -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-prof --cpu-videos 0 --exact-steps 0 --fast-steps 0 --recall-steps 0 --legs none > $O/p$i.log 2>&1
done
F="$F" python3 - <<'PY'
import csv, glob, collections, os
flt = os.environ["F"].split("|")
for one in flt:
    tot = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob("/tmp/pmc_ek/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if one in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print("==", one)
    for k in sorted(tot):
        print(f"{k:32s} {tot[k] / max(n[k], 1):16.0f}  (avg over {n[k]} dispatch rows)")
PY
