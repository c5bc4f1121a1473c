import sys, os
sys.path.insert(0, "/root/repo")
import torch
from pets_face_recognition_amd._hip import lib, ops as o
dev = "cuda:0"
lib.pfr_set_tuning(b"sconv", 2)
for N in (16, 32, 64, 128, 256, 512):
    x = torch.randn(N, 56, 56, 64, device=dev).bfloat16()
    w = (torch.randn(64, 3, 3, 64, device=dev) / 24).bfloat16()
    y = torch.empty_like(x)
    for _ in range(3):
        o.conv2d_fwd(x, w, stride=1, pad=1, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        o.conv2d_fwd(x, w, stride=1, pad=1, out=y)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nblk = N * 98
    print(f"N={N:4d}: {us:7.1f} us  patches/wave {nblk / 1024:6.1f}  us per patch-round {us / (nblk / 1024):6.3f}  {2 * N * 3136 * 64 * 576 / us / 1e6:6.0f} TFLOP/s")
