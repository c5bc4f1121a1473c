import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pets_face_recognition_amd._hip import ops, lib
from pets_face_recognition_amd._hip.lib import LIB_PATH
dll = ctypes.CDLL(LIB_PATH)
CASES = {'c3x3_256_h14': (256, 14, 256, 256, 3), 'c1x1_1024_256_h14': (256, 14, 1024, 256, 1), 'c1x1_256_1024_h14': (256, 14, 256, 1024, 1),
         'c3x3_128_h28': (256, 28, 128, 128, 3), 'c1x1_64_256_h56': (256, 56, 64, 256, 1)}
lib.pfr_set_tuning(b"igemm_ws", 2)
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device='cuda')
for name, (N, H, C, Co, R) in CASES.items():
    x = torch.randn(N, H, H, C, device='cuda').bfloat16()
    w = (torch.randn(Co, R, R, C, device='cuda') / (C * R * R) ** 0.5).bfloat16()
    fl = 2.0 * N * H * H * Co * R * R * C
    out = []
    for dbg, label in ((0, "full"), (2, "no stores"), (16, "no DMA"), (32, "no MFMA"), (96, "no MFMA/ds_read"), (18, "no DMA, no stores"), (50, "only barriers+ds_read")):
        dll.pfr_debug_igemm_flags(dbg)
        y, part = ops.conv2d_fwd(x, w, stride=1, pad=(R - 1) // 2, stats=False)
        t = 0.0
        for _ in range(3):
            big.add_(1.0)
            a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            a.record(); ops.conv2d_fwd(x, w, stride=1, pad=(R - 1) // 2, stats=False, out=y); e.record()
            torch.cuda.synchronize(); t += a.elapsed_time(e) / 3
        out.append(f"{label}: {t*1e3:.1f}us")
    print(name, f"({fl/1e9:.1f} GF; MFMA floor {fl/2.5e15*1e6:.0f}us)", " | ".join(out), flush=True)
