#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# last step: find the last s2d_input kernel
idx = max(i for i, n in enumerate(names) if 's2d_input' in n)
prev = max(i for i, n in enumerate(names[:idx]) if 's2d_input' in n)
step = rows[prev:idx]
t0 = int(step[0]['Start_Timestamp'])
out = []
last_end = t0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.append(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:7.1f}  gap {(s-last_end)/1e3:6.1f}  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:60]}")
    last_end = max(last_end, e)
print("\n".join(out))
PY
