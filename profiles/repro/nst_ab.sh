#!/bin/bash
# wgrad3 LDS ring depth: 4 stages (64 KB, two workgroups per CU; default) vs 3 (48 KB, three per CU): csrc/build_ab/libpfr_hip_nst3.so
for rep in 1 2 3; do
  for v in default nst3; do
    L=""; [ $v = nst3 ] && L=$GRAFT_REPO_ROOT/pets-face-recognition_amd/csrc/build_ab/libpfr_hip_nst3.so
    PFR_LIB_PATH=$L timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], 'wgrad', d['roofline']['by_entry_point_ms'].get('pfr_conv2d_wgrad'))"
  done
done
