#!/bin/bash
# rows in flight per thread in bn_bwd_reduce (PFR_BNR_ROWS = 2 / 4 (default) / 8) and workgroups per launch (PFR_BN_TARGET)
run() { PFR_LIB_PATH=$1 PFR_BN_TARGET=$2 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['roofline']['by_entry_point_ms']; print('$3', d['ms_per_step'], 'reduce', e['pfr_bn_bwd_reduce'], 'apply', e['pfr_bn_bwd_apply'], 'act_mask', e['pfr_bn_act_mask'], 'act', e['pfr_bn_act'])"; }
B=$GRAFT_REPO_ROOT/pets-face-recognition_amd/csrc/build_ab
run "" "" rows4
run $B/libpfr_hip_bnr2.so "" rows2
run $B/libpfr_hip_bnr8.so "" rows8
run "" 256 rows4_t256
run "" 1024 rows4_t1024
run $B/libpfr_hip_bnr8.so 256 rows8_t256
run $B/libpfr_hip_bnr8.so 1024 rows8_t1024
