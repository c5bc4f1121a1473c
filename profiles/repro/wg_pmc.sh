#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for dbg in 0 4; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pm
  PFR_WGRAD_DBG=$dbg rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm -o pm -- python $R/profiles/repro/wg_one.py a > /tmp/pm.log 2>&1
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  python - "$f" dbg$dbg <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r['Kernel_Name'][:40]
    if 'wgrad2' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in acc.items():
    n = sum(1 for r in rows if r['Kernel_Name'][:40] == k and r['Counter_Name'] == list(d)[0])
    print(sys.argv[2], k, 'dispatches', n, {c: round(v / n) for c, v in d.items()})
PY
done; done
