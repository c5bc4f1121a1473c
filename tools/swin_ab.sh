#!/bin/bash
# Interleaved Swin-T bs 128 step A/B on one box: tools/swin_ab.sh "<env A>" "<env B>" [repeats]
A="$1"; B="$2"; REP="${3:-3}"
cd "$(dirname "$0")/.."
for r in $(seq 1 $REP); do
  for cfg in "$A" "$B"; do
    out=$(env $cfg python bench.py --arch swin_t --batch 128 --steps 40 --warmup 10 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
    echo "[$cfg] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["value"], "img/s", "loss", d["config"]["loss"])')"
  done
done
