#!/bin/bash
# GPU idle time between the optimiser kernel of step i and the first kernel of step i+1 over a long run (is the host ahead?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/sg
rocprofv3 --kernel-trace --output-format csv -d /tmp/sg -o t -- python $R/bench.py $@ --steps 30 --warmup 5 --no-extra --no-cpu-baseline --no-roofline > /tmp/sg.log 2>&1
f=$(find /tmp/sg -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
gaps = []
for i, r in enumerate(rows[:-1]):
    if 'sgd_kernel' in r['Kernel_Name'] and 'sgd_kernel' not in rows[i + 1]['Kernel_Name']:
        gaps.append((int(rows[i + 1]['Start_Timestamp']) - int(r['End_Timestamp'])) / 1e3)
print('gaps after the last sgd kernel of each step (us):', [round(g) for g in gaps])
PY
