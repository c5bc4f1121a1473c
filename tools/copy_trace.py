"""Which host code issues the device-to-device copies of a training step: aten::copy_ / clone under torch.profiler, grouped by the
innermost repo frame.   python tools/copy_trace.py"""
import argparse, collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

args = argparse.Namespace(arch="resnet50", batch=256, classes=10000, dtype="bf16")
dev = torch.device("cuda", 0)
ml, opt = bench.build(args, dev)
x = torch.rand(args.batch, 3, 224, 224, device=dev)
y = torch.randint(0, args.classes, (args.batch,), device=dev)


def step():
    opt.zero_grad()
    out = ml(x, y)
    out["loss"].backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::zero_", "aten::fill_", "aten::add_", "aten::mul_", "aten::to", "aten::_to_copy"):
        frames = [f for f in (e.stack or []) if "repo" in f or "bench" in f]
        cnt[(e.name, frames[0] if frames else (e.stack[0] if e.stack else "?"))] += 1
for k, v in cnt.most_common(30):
    print(v, k)
names = collections.Counter(e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
print([(k, v) for k, v in names.most_common(60) if "copy" in k.lower() or "elementwise" in k.lower() or "fill" in k.lower()])
