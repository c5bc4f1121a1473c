#!/bin/bash
# kernel-trace time of wgrad9_kernel under the experiment switches (development aid): tools/w9_exp.sh "0 2 8 15"
export TMPDIR=/tmp WGRAD_AB=wgrad9
for e in $1; do
  rm -rf /tmp/w9x; ( cd /tmp && PFR_W9_EXP=$e rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/w9x -o st -- python $OLDPWD/tools/wgrad_bench.py resnet > /dev/null 2>&1 )
  f=$(find /tmp/w9x -name "*kernel_stats.csv" | head -1)
  echo "EXP $e: $(grep wgrad9 $f | sed 's/"void wgrad9_kernel<\([0-9]*\)>(Wg9Params)",[0-9]*,[0-9]*,\([0-9.]*\),.*/<\1> \2 ns/' | sort | tr '\n' ' ')"
done
