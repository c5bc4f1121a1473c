"""Which torch (aten) operators run inside one train step besides the C-ABI launches — they show up as __amd_rocclr_copyBuffer /
FillFunctor kernels in the rocprof tables.  Prints operator counts with the python frames that issued them.
usage: python profiles/repro/torch_ops.py [resnet50|swin_t] [batch]"""
import sys, os, types, collections, importlib.util, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else (256 if arch == "resnet50" else 128)
args = types.SimpleNamespace(arch=arch, dtype="bf16", classes=10000, batch=batch)
dev = torch.device("cuda:0")
ml, opt = bench.build(args, dev)
g = torch.Generator(device="cpu").manual_seed(123)
x = torch.rand(batch, 3, 224, 224, generator=g).to(dev)
y = torch.randint(0, 10000, (batch,), generator=g).to(dev)


def step():
    opt.zero_grad()
    out = ml(x, y)
    out["loss"].backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.cpu_parent is None or (ev.cpu_parent is not None and not ev.cpu_parent.name.startswith("aten::") and ev.name.startswith("aten::")):
        st = [f for f in (ev.stack or []) if "site-packages/torch" not in f and "<built-in" not in f][:2]
        cnt[(ev.name, str(ev.input_shapes)[:60], " <- ".join(s.split("/")[-1] for s in st))] += 1
for (name, shp, where), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:4d}  {name:28s} {shp:60s} {where}")
