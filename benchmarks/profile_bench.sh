# Profiles of the default benchmark on the MI355X box (run from the repo root through gpurun):
#   TAG=r02a bash benchmarks/profile_bench.sh
# Raw rocprofv3 output goes to /tmp (too large to travel back); the summaries of benchmarks/pmc_summary.py and the
# per-kernel stats land in gpurun_out/summary_$TAG/ -- copy them into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r03a}
O=/tmp/prof_$TAG
S=$R/gpurun_out/summary_$TAG
rm -rf $O $S; mkdir -p $O $S
cd $R
HEAD="--cpu-videos 0 --exact-steps 0 --fast-steps 0 --recall-steps 0 --legs none"
# (1) per-kernel durations of the headline workload (13 forward passes), and of the whole default command with its extra legs
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py $HEAD > $O/stats_stdout.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_full -o bench -- python3 bench.py --cpu-videos 0 --exact-steps 0 --fast-steps 0 --recall-steps 0 > $O/stats_full_stdout.log 2>&1
# (2) HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md), then two SQ passes
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-prof $HEAD > $O/pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d $O/pmc_sq1 -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-prof $HEAD > $O/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_sq2 -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-prof $HEAD > $O/pmc_sq2.log 2>&1
# (3) HBM traffic of the training step (8 and 64 videos), same two separate passes
for v in 8 64; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_train${v}_$c -o pmc -- python3 bench.py --mode train --train-videos $v --steps 3 --warmup 2 --no-prof > $O/pmc_train${v}_$c.log 2>&1
  done
done
python3 benchmarks/pmc_summary.py $O $S $TAG > $S/${TAG}_pmc_summary.json 2> $S/pmc_summary.err
cp $O/stats_full/bench_kernel_stats.csv $S/${TAG}_bench_full_kernel_stats.csv 2>/dev/null
grep "^{" $O/stats_stdout.log | tail -1 > $S/${TAG}_bench_under_rocprof.json.log
ls -la $S
du -sh $O
