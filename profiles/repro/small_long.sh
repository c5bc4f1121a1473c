#!/bin/bash
# launches with few workgroups but long durations (latency chains) in a train step; arg: extra bench flags
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/sl
PFR_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/sl -o t -- python $R/bench.py $@ --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline > /tmp/sl.log 2>&1
f=$(find /tmp/sl -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    wg = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
    nwg = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(wg, 1)
    acc[(r['Kernel_Name'][:60], nwg)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
rows = []
for k, v in acc.items():
    v = sorted(v)
    rows.append((len(v) / 7.0 * v[len(v) // 2], k, len(v), v[len(v) // 2]))
for tot, k, n, med in sorted(rows, reverse=True):
    if k[1] <= 1024 and med > 6.0: print('%8.1f us/step  %-60s wgs %6d  n %4d  median %7.1f us' % (tot, k[0], k[1], n, med))
PY
