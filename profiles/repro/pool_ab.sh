#!/bin/bash
for v in 32 64; do
  PFR_POOL_LAG=$v python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('resnet lag=$v', d['value'], d['ms_per_step'])"
done
for v in 48 96; do
  PFR_POOL_DEPTH=$v python bench.py --arch swin_t --batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('swin depth=$v', d['value'], d['ms_per_step'])"
done
