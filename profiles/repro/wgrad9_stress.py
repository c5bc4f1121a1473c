"""Race screen of pfr_wgrad9.hip (new synchronisation structure: early stage hand-over, staggered LDS-DMA issue, rolling fragment
pipeline) at the bs-256 geometries: every launch is repeated with unrelated HBM traffic and a concurrent convolution on a second stream
and compared BIT FOR BIT with the first result, which is itself checked against the tile kernel.
python profiles/repro/wgrad9_stress.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pets_face_recognition_amd._hip import lib, ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
noise = torch.empty(256 * 1024 * 1024, dtype=torch.bfloat16, device=dev)
nx = torch.randn(256, 28, 28, 128, device=dev, generator=g).bfloat16()
nw = torch.randn(128, 3, 3, 128, device=dev, generator=g).bfloat16() / 34
side = torch.cuda.Stream()
bad = 0
for (N, H, C, Co) in ((256, 56, 64, 64), (256, 28, 128, 128), (256, 14, 256, 256), (256, 7, 512, 512), (96, 14, 256, 512)):
    x = torch.randn(N, H, H, C, device=dev, generator=g).bfloat16()
    dy = torch.randn(N, H, H, Co, device=dev, generator=g).bfloat16()
    lib.pfr_set_tuning(b"wgrad9", 0)
    base = ops.conv2d_wgrad(x, dy, 3, 3, 1, 1)
    lib.pfr_set_tuning(b"wgrad9", 2)
    ref = None
    for it in range(reps):
        with torch.cuda.stream(side):
            noise.add_(1)
            ops.conv2d_fwd(nx, nw, stride=1, pad=1)
        out = ops.conv2d_wgrad(x, dy, 3, 3, 1, 1)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
            err = ((ref - base).abs().max() / base.abs().max()).item()
            assert err < 1e-5, err
        elif not torch.equal(out, ref):
            bad += 1
            print(f"MISMATCH {N}x{H}x{H} C{C}->{Co} repetition {it}: max diff {(out - ref).abs().max().item():.3e}")
    print(f"{H}x{H} C{C}->{Co}: {reps} repetitions under load, bit-identical" if not bad else f"{H}x{H}: {bad} mismatches")
lib.pfr_set_tuning(b"wgrad9", 1)
print("OK" if bad == 0 else "FAILED")
sys.exit(1 if bad else 0)
