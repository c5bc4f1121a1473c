#!/bin/bash
# where do torch's own tiny launches (copyBuffer, FillFunctor, elementwise) sit in a train step?
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
marks = [i for i, n in enumerate(names) if 's2d_input' in n]
a, b = marks[-3], marks[-2]
step = rows[a:b]
ctx = collections.Counter()
for i, r in enumerate(step):
    n = r['Kernel_Name']
    if 'copyBuffer' in n or 'FillFunctor' in n or 'at::native' in n:
        prev = step[i - 1]['Kernel_Name'][:40] if i else ''
        nxt = step[i + 1]['Kernel_Name'][:40] if i + 1 < len(step) else ''
        ctx[(n[:60], r.get('Queue_Id'), prev, nxt)] += 1
for k, v in sorted(ctx.items(), key=lambda kv: -kv[1])[:40]:
    print(v, k)
PY
