#!/bin/bash
for s in 0 3 4 5 6 7 8 10 12 14 16 20 28; do
  echo "splits=$s $(PFR_WGRAD_FORCE_SPLITS=$s timeout 100 python tools/wgrad_micro.py 2>&1 | grep -v n32 | awk '{printf "%s %s/%s | ", $1, $3, $10}')"
done
