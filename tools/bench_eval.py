"""Embedding-extraction throughput (Controller.validation_step / test_step path: eval-mode backbone, no labels):
   python tools/bench_eval.py [arch] [batch] [steps]"""
import sys, os, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda", 0)
ml, _ = bench.build(types.SimpleNamespace(arch=arch, dtype="bf16", classes=10000, batch=batch), dev)
ml.eval()
x = torch.rand(batch, 3, 224, 224).to(dev)
with torch.no_grad():
    for _ in range(5):
        emb = ml(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        emb = ml(x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"{arch} eval bs={batch}: {batch*steps/dt:.0f} img/s, {dt/steps*1e3:.2f} ms/batch, emb {tuple(emb.shape)} finite={bool(torch.isfinite(emb.float()).all())}")
