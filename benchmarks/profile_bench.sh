cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r01m
rm -rf $O; mkdir -p $O
cd $R
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --cpu-videos 0 --exact-steps 0 > $O/stats_stdout.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python3 bench.py --steps 2 --warmup 1 --cpu-videos 0 --no-prof --exact-steps 0 > $O/pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d $O/pmc_sq1 -o pmc -- python3 bench.py --steps 2 --warmup 1 --cpu-videos 0 --no-prof --exact-steps 0 > $O/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_sq2 -o pmc -- python3 bench.py --steps 2 --warmup 1 --cpu-videos 0 --no-prof --exact-steps 0 > $O/pmc_sq2.log 2>&1
ls -R $O | head -40
du -sh $O
