#!/bin/bash
# Collects the evidence bench.py's roofline block refers to (run on the GPU box from the repo root):
#   tools/collect_profiles.sh <tag>      e.g.  tools/collect_profiles.sh r01_d
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/<tag>_kernel_stats.csv
# 2. two PMC passes (the profiled command runs 2 warm-up + 5 timed + 3 host-timing steps = 10 steps) (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains) -> gpurun_out/<tag>_traffic.json
# 3. the bench lines themselves (default = ResNet-50 bs 256; Swin-T bs 128)          -> gpurun_out/<tag>_bench_*.json
set -u
TAG=${1:-r01_x}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
# Profiling passes run with the weight-gradient side stream OFF (PFR_SIDE_STREAM=0): that is the condition under which
# bench.py takes its per-launch HIP-event timings (a tracer forces serial execution), so the per-kernel averages of the
# two agree; with the side stream on, concurrently running kernels stretch each other's durations.
export PFR_SIDE_STREAM=0
CMD="python $PWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-extras"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o stats -- $CMD > $OUT/${TAG}_prof_run.log 2>&1 )
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/${TAG}_kernel_stats_bench_resnet50_bs256_bf16.csv
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${TAG}_$c -o pmc -- $CMD > $OUT/${TAG}_pmc_$c.log 2>&1 )
done
python tools/hbm_traffic.py $OUT/pmc_${TAG}_FETCH_SIZE $OUT/pmc_${TAG}_WRITE_SIZE 10 > $OUT/${TAG}_traffic.json
# one SQ pass (its own run: never together with the TCC passes or another trace domain): where the matrix pipe waits, for the five longest kernels
SQSET="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
( cd /tmp && rocprofv3 --kernel-trace --pmc $SQSET --output-format csv -d $OUT/pmc_${TAG}_SQ -o pmc -- $CMD > $OUT/${TAG}_pmc_SQ.log 2>&1 )
python tools/pmc_top5.py $OUT/pmc_${TAG}_SQ 5 > $OUT/${TAG}_pmc_top5.txt 2>&1
rm -rf $OUT/pmc_${TAG}_SQ
# per-geometry table of the conv launches against max(MFMA, HBM) bounds (serialised launches, HIP events)
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --detail $OUT/${TAG}_detail.json > /dev/null 2>&1
python tools/layer_roofline.py $OUT/${TAG}_detail.json $OUT/${TAG}_layer_roofline.md > /dev/null
# Swin-T bs 128 (BASELINE config 4): kernel stats + HBM traffic of its conv/linear/attention family
SCMD="python $PWD/bench.py --arch swin_t --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-extras"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_swin -o stats -- $SCMD > $OUT/${TAG}_prof_swin_run.log 2>&1 )
f=$(find $OUT/prof_${TAG}_swin -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/${TAG}_kernel_stats_bench_swin_t_bs128_bf16.csv
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${TAG}_swin_$c -o pmc -- $SCMD > $OUT/${TAG}_pmc_swin_$c.log 2>&1 )
done
python tools/hbm_traffic.py $OUT/pmc_${TAG}_swin_FETCH_SIZE $OUT/pmc_${TAG}_swin_WRITE_SIZE 10 "swin_t bs128 bf16 224x224 train step" > $OUT/${TAG}_traffic_swin.json
( cd /tmp && rocprofv3 --kernel-trace --pmc $SQSET --output-format csv -d $OUT/pmc_${TAG}_swin_SQ -o pmc -- $SCMD > $OUT/${TAG}_pmc_swin_SQ.log 2>&1 )
python tools/pmc_top5.py $OUT/pmc_${TAG}_swin_SQ 5 > $OUT/${TAG}_pmc_top5_swin.txt 2>&1
rm -rf $OUT/pmc_${TAG}_swin_SQ
# 10k x 1M x 512 match (BASELINE config 5): kernel stats
MCMD="python $PWD/tools/bench_match.py"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_match -o stats -- $MCMD > $OUT/${TAG}_bench_match.json 2> $OUT/${TAG}_prof_match_run.log )
f=$(find $OUT/prof_${TAG}_match -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/${TAG}_kernel_stats_match_10kx1M.csv
rm -rf $OUT/prof_${TAG}_swin $OUT/prof_${TAG}_match $OUT/pmc_${TAG}_swin_FETCH_SIZE $OUT/pmc_${TAG}_swin_WRITE_SIZE
unset PFR_SIDE_STREAM
python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench_default.json
python bench.py --arch swin_t --batch 128 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_swin_t_bs128.json
rm -rf $OUT/prof_$TAG $OUT/pmc_${TAG}_FETCH_SIZE $OUT/pmc_${TAG}_WRITE_SIZE
ls -la $OUT | tail -12
