#!/bin/bash
# side-stream hand-over granularity: one fork per n weight gradients (PFR_WGRAD_BATCH=n), headline step time
for rep in 1 2; do
  for v in 1 2 4 8 16 64; do
    PFR_WGRAD_BATCH=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('WGRAD_BATCH=$v', d['value'], d['ms_per_step'])"
  done
done
