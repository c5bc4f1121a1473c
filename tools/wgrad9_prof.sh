#!/bin/bash
# kernel times + SQ counters of the 3x3 weight-gradient launches (run on the GPU box from the repo root): tools/wgrad9_prof.sh <tag>
TAG=${1:-w9}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp WGRAD_AB=wgrad9
CMD="python $PWD/tools/wgrad_bench.py resnet"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o st -- $CMD > $OUT/${TAG}_run.log 2>&1 )
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${TAG}_kernel_stats.csv
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_${TAG}_$n -o pmc -- $CMD > /dev/null 2>&1 )
done
python - <<P
import csv, glob, collections
out = "$OUT"; tag = "$TAG"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f"{out}/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "wgrad" not in k: continue
        key = (k, r["Grid_Size"], r.get("LDS_Block_Size", ""))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(key, r["Counter_Name"])] += 1
with open(f"{out}/{tag}_pmc.txt", "w") as fo:
    for key, d in sorted(agg.items()):
        fo.write(str(key) + " " + str({c: round(v / cnt[(key, c)]) for c, v in sorted(d.items())}) + "\n")
print(open(f"{out}/{tag}_pmc.txt").read())
P
grep -i "wgrad" $OUT/${TAG}_kernel_stats.csv | cut -c1-200
tail -5 $OUT/${TAG}_run.log
rm -rf $OUT/prof_$TAG $OUT/pmc_${TAG}_*
