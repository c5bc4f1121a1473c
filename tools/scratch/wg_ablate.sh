#!/bin/bash
for ring in 0 4; do for dbg in 0 1 2 4 3; do
  echo "ring=$ring dbg=$dbg"; PFR_WGRAD_RING=$ring PFR_WGRAD_DBG=$dbg timeout 200 python tools/wgrad_micro.py 2>&1 | grep "w3x3_256_h14 \|w3x3_64_h56 \|w1x1_256_1024\|w1x1_64_256" | cut -c1-60
done; done
