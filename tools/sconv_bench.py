"""A/B of the streaming 1x1 kernel (pfr_sconv.hip) against the tile kernels on the ResNet-50 bs-256 geometries it takes.
usage: python tools/sconv_bench.py [--check]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pets_face_recognition_amd._hip import lib, ops as o

GEOMS = [  # (H, C, Cout, stride, launches per step)
    (56, 256, 64, 1, 6), (56, 64, 256, 1, 4), (14, 256, 1024, 1, 6), (28, 128, 512, 1, 4), (28, 512, 128, 1, 7),
    (56, 256, 128, 1, 1), (56, 128, 256, 1, 1), (56, 64, 64, 1, 2), (56, 256, 512, 2, 1), (28, 256, 512, 1, 1),
    (28, 512, 256, 1, 1), (28, 512, 1024, 2, 1), (14, 512, 1024, 1, 1),
]
dev = "cuda:0"
STATS = "--nostats" not in sys.argv
N = 256
tot = [0.0, 0.0]
for H, C, Co, sd, cnt in GEOMS:
    x = torch.randn(N, H, H, C, device=dev).bfloat16()
    w = (torch.randn(Co, 1, 1, C, device=dev) / C ** 0.5).bfloat16()
    OH = H // sd
    y = torch.empty(N, OH, OH, Co, device=dev, dtype=torch.bfloat16)
    M = N * OH * OH
    res = []
    ys = []
    for mode in (0, 2):
        lib.pfr_set_tuning(b"sconv", mode)
        mt = lib.pfr_conv2d_mtile(N, H, H, C, Co, 1, 1, sd, 0, OH, OH, 1, 1, 0)
        part = torch.empty(((M + mt - 1) // mt, 2, Co), device=dev, dtype=torch.float32)
        for _ in range(3):
            o.conv2d_fwd(x, w, stride=sd, out=y, stats=STATS, stats_buf=part if STATS else None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            o.conv2d_fwd(x, w, stride=sd, out=y, stats=STATS, stats_buf=part if STATS else None)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
        ys.append(y.clone())
    lib.pfr_set_tuning(b"sconv", 1)
    nbytes = (M * sd * sd * 0 + N * OH * OH * C * 2) + M * Co * 2 + Co * C * 2
    same = torch.equal(ys[0], ys[1])
    tot[0] += res[0] * cnt; tot[1] += res[1] * cnt
    print(f"{H:3d}^2 {C:4d}->{Co:4d} s{sd}: tile {res[0]:7.1f} us  sconv {res[1]:7.1f} us  ({nbytes / res[1] / 1e6:5.2f} TB/s, HBM floor @6.3 {nbytes / 6.3e6:6.1f} us)  x{cnt}  identical={same}")
print(f"per step: tile {tot[0] / 1e3:.3f} ms, sconv {tot[1] / 1e3:.3f} ms")

# residual-join data gradients (pfr_conv2d_dgrad_join) of the identity blocks
print("dgrad_join:")
tot = [0.0, 0.0]
st = torch.cuda.current_stream().cuda_stream
for H, C, Co, cnt in [(56, 64, 256, 2), (28, 128, 512, 3), (14, 256, 1024, 5)]:
    dy = torch.randn(N, H, H, C, device=dev).bfloat16()
    wt = (torch.randn(Co, 1, 1, C, device=dev) / C ** 0.5).bfloat16()
    rs = torch.randn(N, H, H, Co, device=dev).bfloat16()
    mk = torch.randint(0, 256, (N * H * H, Co // 8), device=dev, dtype=torch.uint8)
    dx = torch.empty(N, H, H, Co, device=dev, dtype=torch.bfloat16)
    res, ys = [], []
    for mode in (0, 2):
        lib.pfr_set_tuning(b"sconv", mode)
        call = lambda: lib.pfr_conv2d_dgrad_join(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), 1, N, H, H, C, Co, 1, 1, 0, 0, H, H, rs.data_ptr(), mk.data_ptr(), st)
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
        ys.append(dx.clone())
    lib.pfr_set_tuning(b"sconv", 1)
    M = N * H * H
    nbytes = M * C * 2 + 2 * M * Co * 2 + M * Co // 8
    tot[0] += res[0] * cnt; tot[1] += res[1] * cnt
    print(f"{H:3d}^2 {C:4d}->{Co:4d}: tile {res[0]:7.1f} us  sconv {res[1]:7.1f} us  ({nbytes / res[1] / 1e6:5.2f} TB/s, HBM floor @6.3 {nbytes / 6.3e6:6.1f} us)  x{cnt}  identical={torch.equal(ys[0], ys[1])}")
print(f"per step: tile {tot[0] / 1e3:.3f} ms, sconv {tot[1] / 1e3:.3f} ms")

# 3x3 / 64 -> 64 channels at 56x56 (layer1 conv2 forward and data gradient): halo-staged kernel vs the tile kernel
print("3x3 64->64 @56:")
x = torch.randn(N, 56, 56, 64, device=dev).bfloat16()
w = (torch.randn(64, 3, 3, 64, device=dev) / 24).bfloat16()
y = torch.empty(N, 56, 56, 64, device=dev, dtype=torch.bfloat16)
for stats in (True, False):
    res, ys = [], []
    for mode in (0, 2):
        lib.pfr_set_tuning(b"sconv", mode)
        mt = lib.pfr_conv2d_mtile(N, 56, 56, 64, 64, 3, 3, 1, 1, 56, 56, 1, 1, 0)
        M = N * 56 * 56
        part = torch.empty(((M + mt - 1) // mt, 2, 64), device=dev, dtype=torch.float32)
        for _ in range(3):
            o.conv2d_fwd(x, w, stride=1, pad=1, out=y, stats=stats, stats_buf=part if stats else None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            o.conv2d_fwd(x, w, stride=1, pad=1, out=y, stats=stats, stats_buf=part if stats else None)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
        ys.append(y.clone())
    lib.pfr_set_tuning(b"sconv", 1)
    nbytes = 2 * M * 128
    print(f" stats={stats}: tile {res[0]:7.1f} us  sconv3 {res[1]:7.1f} us  ({nbytes / res[1] / 1e6:5.2f} TB/s, {2 * M * 64 * 576 / res[1] / 1e6:6.0f} TFLOP/s, HBM floor @6.3 {nbytes / 6.3e6:5.1f} us)  identical={torch.equal(ys[0], ys[1])}")
