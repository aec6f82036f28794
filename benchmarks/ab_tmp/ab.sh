B="python bench.py --cpu-videos 0 --exact-steps 0 --fast-steps 0 --recall-steps 0 --legs none --steps 20 --warmup 5"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["resident_packed_batch"]["ms_per_step"], d["roofline"]["frac"], d["oracle_parity"]["identical_triplet_rows"] if d.get("oracle_parity") else None)'
for i in 1 2; do $B 2>&1 | grep "^{" | tail -1 | python -c "$P" new; BIGC_LN_RES=0 $B 2>&1 | grep "^{" | tail -1 | python -c "$P" old; done
