"""How does the 256x256 tile kernel's launch time scale with the number of busy CUs?  (round 5: is a k-split remainder
worth building — does a launch with 256 tiles take the time of one with 196?)   python tools/tile_scaling.py [reps]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
GEOS = {'c3x3_256_h14': (14, 256, 256, 3, 1), 'c1x1_1024_256_h14': (14, 1024, 256, 1, 0), 'c3x3_512_h7': (7, 512, 512, 3, 1),
        'c3x3_128_h28': (28, 128, 128, 3, 1)}
for name, (H, C, Co, R, p) in GEOS.items():
    for N in (64, 128, 192, 256, 300, 334, 400, 512):
        x = torch.randn(N, H, H, C, device='cuda').bfloat16()
        w = (torch.randn(Co, R, R, C, device='cuda') / (C * R * R) ** 0.5).bfloat16()
        y, part = ops.conv2d_fwd(x, w, stride=1, pad=p, stats=True)
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            ops.conv2d_fwd(x, w, stride=1, pad=p, stats=True, out=y, stats_buf=part)
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / reps
        M = N * H * H
        fl = 2.0 * M * Co * R * R * C
        print(f'{name:20s} N {N:4d} M {M:7d} tiles256 {(M + 255) // 256 * ((Co + 255) // 256):5d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s', flush=True)
