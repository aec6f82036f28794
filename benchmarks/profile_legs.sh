# rocprofv3 --kernel-trace --stats of the training step (8 and 64 videos) and of the grounding leg, each alone (run from the repo root through gpurun):
#   TAG=r06z bash benchmarks/profile_legs.sh      -> gpurun_out/summary_$TAG/${TAG}_{train_8,train_64,grounding}_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r06z}
O=/tmp/prof_legs_$TAG
S=$R/gpurun_out/summary_$TAG
rm -rf $O; mkdir -p $O $S
cd $R
for v in 8 64; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_$v -o bench -- python3 bench.py --mode train --train-videos $v --steps 20 --warmup 5 --no-prof > $O/train_$v.log 2>&1
  cp $O/train_$v/bench_kernel_stats.csv $S/${TAG}_train_${v}_kernel_stats.csv 2>/dev/null || cp $(find $O/train_$v -name "*kernel_stats.csv" | head -1) $S/${TAG}_train_${v}_kernel_stats.csv
  grep "^{" $O/train_$v.log | tail -1 > $S/${TAG}_train_${v}_under_rocprof.json.log
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/grd -o bench -- python3 bench.py --videos 8 --cpu-videos 0 --exact-steps 0 --fast-steps 0 --recall-steps 0 --legs grounding --steps 2 --warmup 1 --no-prof > $O/grd.log 2>&1
cp $O/grd/bench_kernel_stats.csv $S/${TAG}_grounding_kernel_stats.csv 2>/dev/null || cp $(find $O/grd -name "*kernel_stats.csv" | head -1) $S/${TAG}_grounding_kernel_stats.csv
ls -la $S | grep -E "train_|grounding"
