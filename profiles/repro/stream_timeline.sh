#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py ${BENCH_ARGS:---arch swin_t --batch 128} --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" ${MARK:-nchw_to_nhwc} <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
marks = [i for i, n in enumerate(names) if (sys.argv[2] if len(sys.argv) > 2 else "nchw_to_nhwc") in n]
# steps are delimited by the input layout kernel: take the segment between the 3rd-last and 2nd-last occurrence groups
starts = [i for k, i in enumerate(marks) if k == 0 or i - marks[k - 1] > 50]
a, b = starts[-3], starts[-2]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in step)
print("step span us", (t1 - t0) / 1e3, "kernels", len(step))
byq = collections.defaultdict(float)
for r in step: byq[r.get('Queue_Id')] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print("busy per queue us", dict(byq))
# idle time of the union of all kernels
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
cur_s, cur_e = iv[0]; busy = 0
for s, e in iv[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("union busy us", busy / 1e3, "idle us", (t1 - t0 - busy) / 1e3)
# main-queue gaps > 3 us
mainq = max(byq, key=byq.get)
last = None; gaps = []; prevn = ''
for r in step:
    if r.get('Queue_Id') != mainq: continue
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if last is not None and s - last > 3000: gaps.append(((s - t0) / 1e3, (s - last) / 1e3, prevn[:34] + " -> " + r['Kernel_Name'][:40]))
    last = e; prevn = r['Kernel_Name']
print("main-queue gaps >3us:", len(gaps), "total", sum(g[1] for g in gaps))
for g in gaps[:60]: print("  at %.0f us gap %.1f before %s" % g)
PY
