#!/bin/bash
# A/B of the decoder on pre-split operands at the headline size (256 videos) and at small batches
for v in 256 64 8; do
for cfg in "BIGC_PRESPLIT=0" "BIGC_PRESPLIT=1" "BIGC_PRESPLIT=1 BIGC_PS_RB3=0"; do
  echo "== videos=$v $cfg"
  env $cfg python bench.py --headline packed --videos $v --legs "" --recall-steps 0 --cpu-videos 0 --exact-steps 0 --fast-steps 0 --steps 10 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('ms_per_step', d['ms_per_step'], 'pairs/s', d['value'], 'gemm TF', r['achieved'], 'frac', r['frac'], 'fam', d['kernel_families_ms_per_step'])
"
done; done
