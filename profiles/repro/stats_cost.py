import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pets_face_recognition_amd._hip import ops
CASES = {'c1x1_64_256_h56': (256, 56, 56, 64, 256, 1, 0), 'c1x1_256_64_h56': (256, 56, 56, 256, 64, 1, 0), 'c1x1_256_1024_h14': (256, 14, 14, 256, 1024, 1, 0),
         'c1x1_128_512_h28': (256, 28, 28, 128, 512, 1, 0), 'c3x3_64_h56': (256, 56, 56, 64, 64, 3, 1), 'c3x3_128_h28': (256, 28, 28, 128, 128, 3, 1)}
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device='cuda')
for name, (N, H, W, C, Co, R, p) in CASES.items():
    x = torch.randn(N, H, W, C, device='cuda').bfloat16()
    w = (torch.randn(Co, R, R, C, device='cuda') / (C * R * R) ** 0.5).bfloat16()
    res = []
    for stats in (True, False):
        y, part = ops.conv2d_fwd(x, w, stride=1, pad=p, stats=stats)
        cold = 0.0
        for _ in range(6):
            big.add_(1.0)
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); ops.conv2d_fwd(x, w, stride=1, pad=p, stats=stats, out=y, stats_buf=part); b.record()
            torch.cuda.synchronize(); cold += a.elapsed_time(b) / 6
        res.append(cold * 1e3)
    print(f"{name:20s} cold us with stats {res[0]:7.1f}  without {res[1]:7.1f}")
