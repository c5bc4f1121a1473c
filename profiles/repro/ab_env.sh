#!/bin/bash
# usage: ab_env.sh VAR v1 v2 ...   -> bench ms/step + entry-point times per value
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); e = d['roofline']['by_entry_point_ms']
        print('$VAR=$v', 'img/s', d['value'], 'ms/step', d['ms_per_step'], 'conv_ms', d['roofline']['conv_ms_per_step'], 'fwd', e['pfr_conv2d_fwd'], 'wgrad', e['pfr_conv2d_wgrad'], 'join', e.get('pfr_conv2d_dgrad_join'), 'frac', d['roofline']['frac'], 'lb', d['roofline']['layer_bound']['frac'])
"
done
