#!/bin/bash
# Interleaved Swin-T bs 128 step comparison of N configurations on one box: tools/swin_abn.sh <repeats> "<env A>" "<env B>" ...   ("-" = defaults)
REP=$1; shift
cd "$(dirname "$0")/.."
for r in $(seq 1 $REP); do
  for cfg in "$@"; do
    e="$cfg"; [ "$cfg" = "-" ] && e="PFR_NOP=1"
    out=$(env $e python bench.py --arch swin_t --batch 128 --steps 40 --warmup 10 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
    echo "[$cfg] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["value"], "img/s")')"
  done
done
