"""Is the GPU idle between two train steps?  Events recorded after opt.step() of step i and before opt.zero_grad() of step i + 1: with the host ahead
of the GPU the two are back to back on the stream (elapsed ~ 0); an idle GPU shows as elapsed > 0.  No profiler attached.   python tools/step_gap.py"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = argparse.Namespace(arch="resnet50", batch=256, classes=10000, dtype="bf16")
dev = torch.device("cuda", 0)
ml, opt = bench.build(args, dev)
x = torch.rand(args.batch, 3, 224, 224, device=dev); y = torch.randint(0, args.classes, (args.batch,), device=dev)
def step():
    opt.zero_grad(); out = ml(x, y); out["loss"].backward(); opt.step()
for _ in range(15): step()
torch.cuda.synchronize()
n = 40
e_end = [torch.cuda.Event(enable_timing=True) for _ in range(n)]; e_start = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
host = []
t0 = time.perf_counter()
for i in range(n):
    e_start[i].record()
    h0 = time.perf_counter(); step(); host.append((time.perf_counter() - h0) * 1e3)
    e_end[i].record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
gaps = [e_end[i].elapsed_time(e_start[i + 1]) * 1e3 for i in range(n - 1)]
steps = [e_start[i].elapsed_time(e_end[i]) for i in range(n)]
print(f"wall {wall:.3f} ms/step; GPU time start->end of a step: median {sorted(steps)[n // 2]:.3f} ms; idle between steps (us): median {sorted(gaps)[len(gaps) // 2]:.1f}, "
      f"min {min(gaps):.1f}, max {max(gaps):.1f}; host enqueue per step: median {sorted(host)[n // 2]:.2f} ms, first 5 {[round(h, 2) for h in host[:5]]}")
