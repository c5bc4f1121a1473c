#!/bin/bash
# kernel timeline of ONE eval-mode forward (BN-folded plan) of ResNet-50 bs 256 bf16: per-kernel durations, gaps, totals by kernel
cd /tmp && export TMPDIR=/tmp
cat > /tmp/eval_one.py <<'PY'
import os, sys, types, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
a = types.SimpleNamespace(arch="resnet50", dtype="bf16", classes=10000, batch=256)
ml, _ = bench.build(a, torch.device("cuda:0"))
ml.eval()
x = torch.rand(int(os.environ.get("EVAL_BS", "256")), 3, 224, 224, device="cuda:0")
with torch.no_grad():
    for _ in range(6):
        e = ml(x)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d /tmp/ke -o ke -- python /tmp/eval_one.py > /tmp/ke.log 2>&1
f=$(find /tmp/ke -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 's2d_input' in n]
step = rows[idx[-2]:idx[-1]]
t0 = int(step[0]['Start_Timestamp'])
last_end = t0
tot = collections.OrderedDict()
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:7.1f}  gap {(s-last_end)/1e3:6.1f}  {r['Kernel_Name'][:70]}  grid {r.get('Grid_Size','?')}")
    k = r['Kernel_Name'][:70]
    tot[k] = tot.get(k, 0) + (e - s)
    last_end = max(last_end, e)
print("---- totals")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{v/1e3:9.1f} us  {k}")
print("span", (last_end - t0) / 1e3, "us; kernels", sum(tot.values()) / 1e3, "us;", len(step), "launches")
PY
