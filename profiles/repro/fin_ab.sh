#!/bin/bash
# A/B of the one-launch BatchNorm backward reduce+finalize (PFR_FUSE_FIN): headline step time, alternating runs
mkdir -p gpurun_out
for rep in 1 2 3; do
  for v in 0 1; do
    PFR_FUSE_FIN=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('FUSE_FIN=$v', d['value'], d['ms_per_step'])"
  done
done
