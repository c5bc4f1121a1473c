#!/bin/bash
# BatchNorm-backward sums out of the streaming kernels' epilogues (PFR_FUSE_BNB=2) vs the separate reduce pass (0)
for rep in 1 2 3; do
  for v in 0 2; do
    PFR_FUSE_BNB=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['roofline']['by_entry_point_ms']; print('FUSE_BNB=$v', d['value'], d['ms_per_step'], 'reduce', e.get('pfr_bn_bwd_reduce'), 'dgrad_bn', e.get('pfr_conv2d_dgrad_bn'), 'join', e.get('pfr_conv2d_dgrad_join'), 'fwd', e.get('pfr_conv2d_fwd'), 'fin', e.get('pfr_bn_bwd_finalize'))"
  done
done
