#!/bin/bash
# VERDICT r5 item 1 (i)+(ii): run on the GPU box from the repo root:   bash tools/tile_contention.sh > gpurun_out/r06_tile_contention.txt 2>&1
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT
T="python $ROOT/tools/tile_contention.py"
TR=$ROOT/pets-face-recognition_amd/csrc/libpfr_hip_trace.so
echo "== 1. launch time vs busy CUs, random vs zero-filled operands, rotated k-loop start (no profiler) =="
for c in c3x3_256_h14 c1x1_1024_256_h14 c3x3_512_h7; do
  $T fwd $c 64,128,192,256,334 data=randn,zeros krot=0
  $T fwd $c 128,256,334 krot=1,5,7
done
echo "== 1b. rotated k-loop: result against the unrotated launch (fp32 summation order differs) =="
python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
from pets_face_recognition_amd._hip import ops, lib
for (H, C, Co, R, p) in ((14, 256, 256, 3, 1), (14, 1024, 256, 1, 0), (7, 512, 512, 3, 1), (14, 256, 1024, 1, 0), (28, 128, 128, 3, 1)):
    x = torch.randn(64, H, H, C, device='cuda').bfloat16(); w = (torch.randn(Co, R, R, C, device='cuda') / (C * R * R) ** 0.5).bfloat16()
    lib.pfr_set_tuning(b"igemm_krot", 0); y0, p0 = ops.conv2d_fwd(x, w, stride=1, pad=p, stats=True)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=p).permute(0, 2, 3, 1)
    for k in (1, 5, 7):
        lib.pfr_set_tuning(b"igemm_krot", k); y1, p1 = ops.conv2d_fwd(x, w, stride=1, pad=p, stats=True)
        print(f"H{H} C{C} Co{Co} R{R} krot {k}: max|y1-y0| {(y1.float()-y0.float()).abs().max().item():.4f}  rel err vs fp32 ref: krot0 {((y0.float()-ref).norm()/ref.norm()).item():.2e} krot {((y1.float()-ref).norm()/ref.norm()).item():.2e}")
PY
echo "== 2. the same tiles on XCDs 0-3 only (trace library: dbg=16) vs all 8 =="
PFR_LIB_PATH=$TR $T fwd c3x3_256_h14 32,64,128 dbg=0
PFR_LIB_PATH=$TR $T fwd c3x3_256_h14 32,64,128 dbg=16
PFR_LIB_PATH=$TR $T fwd c1x1_1024_256_h14 64,128 dbg=0
PFR_LIB_PATH=$TR $T fwd c1x1_1024_256_h14 64,128 dbg=16
echo "== 3. weight gradient wgrad3<128,128>: constant work per workgroup (36 images x 196 rows per split), 72 .. 252 workgroups =="
for ns in "72 2" "108 3" "180 5" "252 7" "288 8" "504 14"; do set -- $ns; $T wgrad c3x3_256_h14 $1 splits=$2; done
for ns in "72 2" "252 7"; do set -- $ns; $T wgrad c3x3_256_h14 $1 splits=$2 data=zeros; done
echo "== 4. smi power / clocks while the launch repeats for 4 s =="
for n in 128 256 334; do
  for d in randn zeros; do
    $T fwd c3x3_256_h14 $n data=$d secs=4 &
    pid=$!
    sleep 2.5
    rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "power|sclk|fclk|mclk" | head -8
    wait $pid
  done
done
echo "== 5. counters (separate passes; 40 dispatches each, first 5 skipped) =="
rocprofv3 -L > $OUT/r06_counters_list.txt 2>&1
SQ="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16"
SQ2="GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC"
TCC="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
TCC2="GRBM_GUI_ACTIVE TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum"
TCP="GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"
pass() {   # name, counters, command...
  local name=$1 ctr=$2; shift 2
  rm -rf $OUT/pmc_$name
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_$name -o pmc -- "$@" > $OUT/pmc_$name.log 2>&1 )
  echo "-- $name: $ctr"
  python $ROOT/tools/pmc_dispatch.py $OUT/pmc_$name "${PMC_FILTER:-igemm_kernel}" 5
  rm -rf $OUT/pmc_$name
}
for n in 128 256 334; do
  pass sq_$n "$SQ" $T fwd c3x3_256_h14 $n reps=40
  pass sq2_$n "$SQ2" $T fwd c3x3_256_h14 $n reps=40
  pass tcc_$n "$TCC" $T fwd c3x3_256_h14 $n reps=40
  pass tcc2_$n "$TCC2" $T fwd c3x3_256_h14 $n reps=40
  pass tcp_$n "$TCP" $T fwd c3x3_256_h14 $n reps=40
done
pass sq_334_zeros "$SQ" $T fwd c3x3_256_h14 334 reps=40 data=zeros
pass sq_334_krot "$SQ" $T fwd c3x3_256_h14 334 reps=40 krot=5
pass tcc_334_krot "$TCC" $T fwd c3x3_256_h14 334 reps=40 krot=5
PMC_FILTER=wgrad3
for ns in "72 2" "252 7"; do set -- $ns
  pass wsq_$1 "$SQ" $T wgrad c3x3_256_h14 $1 splits=$2 reps=40
  pass wtcc_$1 "$TCC" $T wgrad c3x3_256_h14 $1 splits=$2 reps=40
done
