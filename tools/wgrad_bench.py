"""A/B of the weight-gradient tile forms on the ResNet-50 / Swin-T geometries (bs 256 / 128, bf16): 4-wave 128-wide tiles vs the
8-wave 256x256 tile (pfr_set_tuning("wgrad_big", 0 | 2)).  Prints us per launch, TFLOP/s, and the relative difference of the two
results (different split counts -> different fp32 summation order).   python tools/wgrad_bench.py [resnet|swin]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pets_face_recognition_amd._hip import lib, ops  # noqa: E402

RESNET = [  # N, H, W, C, Cout, R, stride, pad
    (256, 14, 14, 256, 256, 3, 1, 1), (256, 28, 28, 128, 128, 3, 1, 1), (256, 7, 7, 512, 512, 3, 1, 1), (256, 56, 56, 64, 64, 3, 1, 1),
    (256, 14, 14, 256, 1024, 1, 1, 0), (256, 14, 14, 1024, 256, 1, 1, 0), (256, 7, 7, 512, 2048, 1, 1, 0), (256, 7, 7, 2048, 512, 1, 1, 0),
    (256, 28, 28, 512, 256, 1, 1, 0), (256, 14, 14, 1024, 512, 1, 1, 0), (256, 28, 28, 256, 256, 3, 2, 1), (256, 14, 14, 512, 512, 3, 2, 1),
    (256, 28, 28, 512, 1024, 1, 2, 0), (256, 14, 14, 1024, 2048, 1, 2, 0), (256, 56, 56, 256, 512, 1, 2, 0),
    (256, 28, 28, 128, 512, 1, 1, 0), (256, 28, 28, 512, 128, 1, 1, 0), (256, 56, 56, 64, 256, 1, 1, 0), (256, 56, 56, 256, 64, 1, 1, 0),
    (256, 56, 56, 64, 64, 1, 1, 0), (256, 56, 56, 256, 128, 1, 1, 0), (256, 28, 28, 256, 512, 1, 1, 0),
]
if os.environ.get("WGRAD_AB") == "wgrad9":
    RESNET = [g for g in RESNET if g[5] == 3 and g[6] == 1]
if os.environ.get("WGRAD_AB") == "swgrad":
    RESNET = [g for g in RESNET if g[5] == 1 and g[6] == 1]
SWIN = [(25088, 1, 1, 1536, 384, 1, 1, 0), (25088, 1, 1, 384, 1536, 1, 1, 0), (25088, 1, 1, 384, 1152, 1, 1, 0), (25088, 1, 1, 384, 384, 1, 1, 0),
        (6272, 1, 1, 3072, 768, 1, 1, 0), (6272, 1, 1, 768, 3072, 1, 1, 0), (6272, 1, 1, 768, 2304, 1, 1, 0), (100352, 1, 1, 768, 192, 1, 1, 0),
        (100352, 1, 1, 192, 768, 1, 1, 0), (401408, 1, 1, 384, 96, 1, 1, 0)]


def run(geom, reps=20):
    N, H, W, C, Co, R, s, p = geom
    OH = (H + 2 * p - R) // s + 1
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(N, H, W, C, device="cuda", generator=g).bfloat16()
    dy = torch.randn(N, OH, OH if W > 1 else 1, Co, device="cuda", generator=g).bfloat16()
    ws = torch.empty(96 << 20, dtype=torch.float32, device="cuda")
    res = []
    key = os.environ.get("WGRAD_AB", "wgrad_big").encode()
    if key == b"swgrad":
        lib.pfr_set_tuning(b"wgrad_big", 0)
    else:
        lib.pfr_set_tuning(b"swgrad", 0)
    for mode in (0, 2):
        lib.pfr_set_tuning(key, mode)
        out = ops.conv2d_wgrad(x, dy, R, R, s, p, workspace=ws)
        for _ in range(3):
            ops.conv2d_wgrad(x, dy, R, R, s, p, out=out, workspace=ws)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            ops.conv2d_wgrad(x, dy, R, R, s, p, out=out, workspace=ws)
        b.record()
        torch.cuda.synchronize()
        res.append((a.elapsed_time(b) / reps * 1e3, out.clone(), lib.pfr_conv2d_wgrad_splits(N * OH * (OH if W > 1 else 1), Co, R * R * C)))
    fl = 2.0 * N * OH * (OH if W > 1 else 1) * Co * R * R * C
    d = ((res[0][1] - res[1][1]).abs().max() / res[0][1].abs().max()).item()
    print(f"{str(geom):44s} base {res[0][0]:7.1f} us {fl / res[0][0] / 1e6:6.0f} TF/s (splits {res[0][2]:3d}) | variant {res[1][0]:7.1f} us "
          f"{fl / res[1][0] / 1e6:6.0f} TF/s (splits {res[1][2]:3d}) | x{res[0][0] / res[1][0]:.2f}  rel diff {d:.1e}", flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "resnet"
    for geom in (SWIN if which == "swin" else RESNET):
        run(geom)
