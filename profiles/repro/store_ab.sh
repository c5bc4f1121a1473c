#!/bin/bash
# A/B: store + EXP_CNT wait as one asm statement (default build) vs builtin store + separate wait (csrc/build_ab/libpfr_hip_splitstore.so)
for rep in 1 2 3; do
  for v in default split; do
    L=""; [ $v = split ] && L=$GRAFT_REPO_ROOT/pets-face-recognition_amd/csrc/build_ab/libpfr_hip_splitstore.so
    PFR_LIB_PATH=$L timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
  done
done
