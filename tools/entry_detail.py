"""Per-launch detail of one C-ABI entry point in a train step (serial, HIP events):
   python tools/entry_detail.py <arch> <batch> <entry> <argidx> [<argidx> ...]   (argidx: positional args that key the table)"""
import sys, os, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pets_face_recognition_amd._hip import set_tracer, EventTracer
arch, batch, entry = sys.argv[1], int(sys.argv[2]), sys.argv[3]
idx = [int(a) for a in sys.argv[4:]]
args = types.SimpleNamespace(arch=arch, dtype="bf16", classes=10000, batch=batch)
dev = torch.device("cuda", 0)
ml, opt = bench.build(args, dev)
x = torch.rand(batch, 3, 224, 224).to(dev); y = torch.randint(0, 10000, (batch,)).to(dev)
def step():
    opt.zero_grad(); out = ml(x, y); out["loss"].backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
tr = EventTracer(); set_tracer(tr); step(); set_tracer(None); torch.cuda.synchronize()
tab = {}
for name, a, e0, e1 in tr.records:
    if name != entry: continue
    k = tuple(a[i] for i in idx)
    o = tab.setdefault(k, [0, 0.0]); o[0] += 1; o[1] += e0.elapsed_time(e1)
for k, (n, ms) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
    print(k, "n=%d" % n, "total %.3f ms" % ms, "avg %.1f us" % (ms / n * 1e3))
