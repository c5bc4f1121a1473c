"""Micro-benchmark of single conv launches (forward kernel) for profiling: python tools/conv_micro.py [reps]"""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
which = sys.argv[2] if len(sys.argv) > 2 else 'all'
CASES = {
    'c3x3_256_h14': (256, 14, 14, 256, 256, 3, 1, 1),
    'c1x1_1024_256_h14': (256, 14, 14, 1024, 256, 1, 1, 0),
    'c1x1_256_1024_h14': (256, 14, 14, 256, 1024, 1, 1, 0),
    'c3x3_64_h56': (256, 56, 56, 64, 64, 3, 1, 1),
    'c1x1_64_256_h56': (256, 56, 56, 64, 256, 1, 1, 0),
    'c3x3_512_h7': (256, 7, 7, 512, 512, 3, 1, 1),
    'c3x3_128_h28': (256, 28, 28, 128, 128, 3, 1, 1),
    'c1x1_512_128_h28': (256, 28, 28, 512, 128, 1, 1, 0),
    'c1x1_128_512_h28': (256, 28, 28, 128, 512, 1, 1, 0),
    'c1x1_256_64_h56': (256, 56, 56, 256, 64, 1, 1, 0),
}
for name, (N, H, W, C, Co, R, s, p) in CASES.items():
    if which != 'all' and which != name: continue
    x = torch.randn(N, H, W, C, device='cuda').bfloat16()
    w = (torch.randn(Co, R, R, C, device='cuda') / (C * R * R) ** 0.5).bfloat16()
    y, part = ops.conv2d_fwd(x, w, stride=s, pad=p, stats=True)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        ops.conv2d_fwd(x, w, stride=s, pad=p, stats=True, out=y, stats_buf=part)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / reps
    # cold variant: evict L2 / MALL between launches by streaming a 1 GiB buffer
    big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device='cuda')
    cold = 0.0
    for _ in range(5):
        big.add_(1.0)
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); ops.conv2d_fwd(x, w, stride=s, pad=p, stats=True, out=y, stats_buf=part); b.record()
        torch.cuda.synchronize(); cold += a.elapsed_time(b) / 5
    del big
    OH = (H + 2 * p - R) // s + 1
    fl = 2.0 * N * OH * OH * Co * R * R * C
    print(f'{name:22s} warm {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s   cold {cold*1e3:8.1f} us {fl/cold/1e9:7.1f} TF/s')
