#!/bin/bash
# Interleaved whole-step comparison of N configurations on one box: tools/step_abn.sh <repeats> <steps> "<env A>" "<env B>" ...   ("-" = no extra environment)
REP=$1; STEPS=$2; shift 2
cd "$(dirname "$0")/.."
for r in $(seq 1 $REP); do
  for cfg in "$@"; do
    e="$cfg"; [ "$cfg" = "-" ] && e="PFR_NOP=1"
    out=$(env $e python bench.py --steps $STEPS --warmup 15 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
    echo "[$cfg] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["value"], "img/s", "loss", d["config"].get("loss"))')"
  done
done
