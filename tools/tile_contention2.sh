#!/bin/bash
# second pass of the contention experiment: the 8-wave 256x256 tile FORCED at every size (the heuristic switches to 128x128 tiles below 160 tiles,
# which the first pass's 98-tile rows silently measured), XCD subset, and the two-stream overlap probe
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT
T="python $ROOT/tools/tile_contention.py"
TR=$ROOT/pets-face-recognition_amd/csrc/libpfr_hip_trace.so
echo "== 1. 256x256 tile forced (tile=4): launch time vs busy CUs, random vs zero-filled =="
$T fwd c3x3_256_h14 32,64,128,192,256,334 data=randn,zeros tile=4
$T fwd c1x1_1024_256_h14 64,128,256,334 data=randn,zeros tile=4
echo "== 2. the same tiles on XCDs 0-3 only (dbg=16) vs all 8, 256x256 tile forced =="
PFR_LIB_PATH=$TR $T fwd c3x3_256_h14 32,64,128 dbg=0 tile=4 data=randn,zeros
PFR_LIB_PATH=$TR $T fwd c3x3_256_h14 32,64,128 dbg=16 tile=4 data=randn,zeros
echo "== 3. two-stream overlap probe =="
python $ROOT/tools/overlap_probe.py 20
echo "== 4. counters at 98 tiles with the 256x256 tile =="
SQ="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"
TCC="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
pass() {
  local name=$1 ctr=$2; shift 2
  rm -rf $OUT/pmc_$name
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$name -o pmc -- "$@" > $OUT/pmc_$name.log 2>&1 )
  echo "-- $name: $ctr"
  python $ROOT/tools/pmc_dispatch.py $OUT/pmc_$name igemm_kernel 5
  rm -rf $OUT/pmc_$name $OUT/pmc_$name.log
}
for n in 64 128; do
  pass sq_t4_$n "$SQ" $T fwd c3x3_256_h14 $n reps=40 tile=4
  pass tcc_t4_$n "$TCC" $T fwd c3x3_256_h14 $n reps=40 tile=4
done
pass sq_t4_256_zeros "$SQ" $T fwd c3x3_256_h14 256 reps=40 tile=4 data=zeros
pass sq_t4_128_zeros "$SQ" $T fwd c3x3_256_h14 128 reps=40 tile=4 data=zeros
