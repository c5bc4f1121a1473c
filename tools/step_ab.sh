#!/bin/bash
# Interleaved whole-step A/B on one box: tools/step_ab.sh "<env assignments A>" "<env assignments B>" [repeats] [steps]
# e.g. tools/step_ab.sh "PFR_TUNING=bnb_tile3=0" "PFR_TUNING=bnb_tile3=1" 3 60   ->  ms_per_step of every run, A and B alternating
A="$1"; B="$2"; REP="${3:-3}"; STEPS="${4:-60}"
cd "$(dirname "$0")/.."
for r in $(seq 1 $REP); do
  for cfg in "$A" "$B"; do
    out=$(env $cfg python bench.py --steps $STEPS --warmup 15 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
    echo "[$cfg] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d["value"], "img/s", "loss", d["config"]["loss"])')"
  done
done
