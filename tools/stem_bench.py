"""Stem convolution (space-to-depth form: 4x4 / stride 1 / pad 2 over [256][112][112][16] -> 64) under the tile-kernel knobs, and its weight gradient.
python tools/stem_bench.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import ops, lib
N, H, C, Co = 256, 112, 16, 64
x = torch.randn(N, H, H, C, device='cuda').bfloat16(); w = (torch.randn(Co, 4, 4, C, device='cuda') / 16).bfloat16()
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device='cuda')
def t(fn, cold):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(6):
        if cold: big.add_(1.0)
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); tot += a.elapsed_time(b)
    return tot / 6 * 1e3
for tile in (-1, 0, 1, 2, 3):
    for p in (1, 2):
        lib.pfr_set_tuning(b"igemm_tile", tile); lib.pfr_set_tuning(b"igemm_p", p)
        try:
            y, part = ops.conv2d_fwd(x, w, stride=1, pad=2, out_hw=(H, H), stats=True)
            fn = lambda: ops.conv2d_fwd(x, w, stride=1, pad=2, out_hw=(H, H), stats=True, out=y, stats_buf=part)
            print(f"fwd tile {tile:2d} igemm_p {p}: warm {t(fn, False):7.1f} us  cold {t(fn, True):7.1f} us", flush=True)
        except Exception as e:
            print(f"fwd tile {tile} igemm_p {p}: {str(e)[:100]}")
lib.pfr_set_tuning(b"igemm_tile", -1); lib.pfr_set_tuning(b"igemm_p", 1)
dy = torch.randn(N, H, H, Co, device='cuda').bfloat16()
ws = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device='cuda')
for wt in (-1, 0, 1, 2, 3):
    for sp in (0, 64, 128, 256, 512):
        lib.pfr_set_tuning(b"wgrad_tile", wt); lib.pfr_set_tuning(b"wgrad_splits", sp)
        try:
            out = ops.conv2d_wgrad(x, dy, 4, 4, 1, 2, workspace=ws)
            fn = lambda: ops.conv2d_wgrad(x, dy, 4, 4, 1, 2, out=out, workspace=ws)
            print(f"wgrad tile {wt:2d} splits {sp:3d}: warm {t(fn, False):7.1f} us  cold {t(fn, True):7.1f} us", flush=True)
        except Exception as e:
            print(f"wgrad tile {wt} splits {sp}: {str(e)[:100]}")
