#!/bin/bash
# Same-box A/B of library builds (every gpurun call lands on another box, and boxes differ by 1-2 % — variants must be compared inside
# one call).  Variant libraries: PFR_BUILD_TAG=<tag> PFR_EXTRA_FLAGS=... csrc/build.sh -> csrc/libpfr_hip_<tag>.so, loaded through
# PFR_LIB_PATH.  Usage: tools/lib_ab.sh "<bench.py args>" rounds tag1 tag2 ...   ("default" = csrc/libpfr_hip.so)
ARGS=$1; ROUNDS=$2; shift 2
C=$PWD/pets-face-recognition_amd/csrc
for r in $(seq $ROUNDS); do
  for t in "$@"; do
    if [ $t = default ]; then L=$C/libpfr_hip.so; else L=$C/libpfr_hip_$t.so; fi
    v=$(PFR_LIB_PATH=$L python bench.py $ARGS --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "round $r $t: $v"
  done
done
