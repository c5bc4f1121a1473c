#!/bin/bash
# per-launch durations of the LayerNorm / attention kernels of a Swin-T step by grid size (side stream off: no concurrency)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/lt
PFR_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o t -- python $R/bench.py --arch swin_t --batch 128 --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-roofline > /tmp/lt.log 2>&1
f=$(find /tmp/lt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list); print(list(csv.DictReader(open(sys.argv[1])).fieldnames))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'layernorm' in n or 'window_attn' in n or 'colsum' in n or 'nchw' in n:
        acc[(n[:40], r.get("Grid_Size_X", r.get("Workgroup_Size_X", "?")))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(acc.items()):
    v = sorted(v)
    print(k, len(v), 'median %.1f us' % v[len(v) // 2], 'min %.1f' % v[0])
PY
