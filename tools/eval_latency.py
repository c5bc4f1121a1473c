"""Eval-mode embedder (BN-folded plan) latency / throughput by batch size — the per-photo loop of generate_tsv.py:233-251 and
Controller.validation_step.  python tools/eval_latency.py [arch]   (PFR_GRAPH_EVAL=1: hipGraph replay of the plan)"""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
a = types.SimpleNamespace(arch=arch, dtype="bf16", classes=10000, batch=256)
dev = torch.device("cuda:0")
ml, _ = bench.build(a, dev)
ml.eval()
for bs in (1, 4, 16, 64, 256):
    x = torch.rand(bs, 3, 224, 224, device=dev)
    with torch.no_grad():
        for _ in range(5):
            ml(x)
        torch.cuda.synchronize()
        n = 50 if bs <= 16 else 20
        t0 = time.perf_counter()
        for _ in range(n):
            e = ml(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        # host time alone: enqueue without waiting
        t1 = time.perf_counter()
        for _ in range(n):
            e = ml(x)
        host = (time.perf_counter() - t1) / n
        torch.cuda.synchronize()
    print(f"{arch} eval bs {bs:4d}: {dt * 1e3:7.3f} ms / batch  {bs / dt:9.1f} img/s   host enqueue {host * 1e3:6.3f} ms", flush=True)
