#!/bin/bash
# streaming 1x1 weight gradient (PFR_SWGRAD: 0 off, 1 narrow-operand shapes, 2 every eligible shape) inside the train step
for rep in 1 2; do
  for v in 0 1 2; do
    PFR_SWGRAD=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['roofline']['by_entry_point_ms']; print('SWGRAD=$v', d['value'], d['ms_per_step'], 'wgrad', e.get('pfr_conv2d_wgrad'))"
  done
done
