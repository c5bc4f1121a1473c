#!/bin/bash
for v in 1 0; do
  PFR_SIDE_STREAM=$v python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('resnet side=$v', d['value'], d['ms_per_step'])"
  PFR_SIDE_STREAM=$v python bench.py --arch swin_t --batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('swin side=$v', d['value'], d['ms_per_step'])"
done
