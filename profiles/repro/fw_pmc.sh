#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in a b c d; do
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  rm -rf /tmp/pm
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm -o pm -- python $R/profiles/repro/fw_one.py $c > /tmp/pm.log 2>&1
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  python - "$f" case_$c <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r['Kernel_Name'][:60]
    if 'igemm' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in acc.items():
    n = sum(1 for r in rows if r['Kernel_Name'][:60] == k and r['Counter_Name'] == list(d)[0])
    dd = {c: round(v / n) for c, v in d.items()}
    print(sys.argv[2], k, 'disp', n, dd, 'VALU/MFMA', round(dd['SQ_INSTS_VALU'] / max(1, dd['SQ_INSTS_MFMA']), 1))
PY
done; done
