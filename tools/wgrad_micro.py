"""Micro-benchmark of weight-gradient launches: python tools/wgrad_micro.py   (cold = L2/MALL flushed between launches)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import ops
CASES = {
    'w3x3_64_h56': (256, 56, 56, 64, 64, 3, 1, 1),
    'w3x3_64_h56_n32': (32, 56, 56, 64, 64, 3, 1, 1),
    'w3x3_128_h28': (256, 28, 28, 128, 128, 3, 1, 1),
    'w3x3_256_h14': (256, 14, 14, 256, 256, 3, 1, 1),
    'w3x3_256_h14_n32': (32, 14, 14, 256, 256, 3, 1, 1),
    'w3x3_512_h7': (256, 7, 7, 512, 512, 3, 1, 1),
    'w1x1_64_256_h56': (256, 56, 56, 64, 256, 1, 1, 0),
    'w1x1_256_1024_h14': (256, 14, 14, 256, 1024, 1, 1, 0),
    'w1x1_1024_256_h14': (256, 14, 14, 1024, 256, 1, 1, 0),
    'w1x1_576_64_h56': (256, 56, 56, 576, 64, 1, 1, 0),
}
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device='cuda')
for name, (N, H, W, C, Co, R, s, p) in CASES.items():
    x = torch.randn(N, H, W, C, device='cuda').bfloat16()
    OH = (H + 2 * p - R) // s + 1
    dy = torch.randn(N, OH, OH, Co, device='cuda').bfloat16()
    ws = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device='cuda')
    out = ops.conv2d_wgrad(x, dy, R, R, s, p, workspace=ws)
    torch.cuda.synchronize()
    reps = 10
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        ops.conv2d_wgrad(x, dy, R, R, s, p, out=out, workspace=ws)
    t1.record(); torch.cuda.synchronize()
    warm = t0.elapsed_time(t1) / reps
    cold = 0.0
    for _ in range(5):
        big.add_(1.0)
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); ops.conv2d_wgrad(x, dy, R, R, s, p, out=out, workspace=ws); b.record()
        torch.cuda.synchronize(); cold += a.elapsed_time(b) / 5
    fl = 2.0 * N * OH * OH * Co * R * R * C
    by = (x.numel() + dy.numel()) * 2
    print(f'{name:20s} warm {warm*1e3:8.1f} us {fl/warm/1e9:7.1f} TF/s | cold {cold*1e3:8.1f} us {fl/cold/1e9:7.1f} TF/s  {by/cold/1e9:6.2f} TB/s (x+dy once)')
