"""Stress test of the streaming kernels at full (bs 256) size: every launch's output (NaN-poisoned beforehand) against the tile
kernel's, repeated — catches rare timing hazards (store-data read-out, counted waits) that small shapes never show.
usage: python profiles/repro/sconv_stress.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pets_face_recognition_amd._hip import lib, ops as o
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = 256
st = torch.cuda.current_stream().cuda_stream
total_bad = 0
for H, C, Co, R, stats, join in [(56, 64, 64, 1, False, False), (56, 64, 64, 1, True, False), (56, 256, 64, 1, False, False), (56, 64, 256, 1, True, False),
                                 (28, 512, 128, 1, False, False), (14, 256, 1024, 1, True, False), (14, 512, 1024, 1, False, False),
                                 (56, 64, 256, 1, False, True), (14, 256, 1024, 1, False, True), (56, 64, 64, 3, True, False), (56, 64, 64, 3, False, False)]:
    x = torch.randn(N, H, H, C, device=dev).bfloat16()
    w = (torch.randn(Co, R, R, C, device=dev) / (C * R * R) ** 0.5).bfloat16()
    res = torch.randn(N, H, H, Co, device=dev).bfloat16() if join else None
    mk = torch.randint(0, 256, (N * H * H, Co // 8), device=dev, dtype=torch.uint8) if join else None
    def run(mode):
        lib.pfr_set_tuning(b"sconv", mode)
        y = torch.full((N, H, H, Co), float("nan"), device=dev, dtype=torch.bfloat16)
        if join:
            lib.pfr_conv2d_dgrad_join(x.data_ptr(), w.data_ptr(), y.data_ptr(), 1, N, H, H, C, Co, 1, 1, 0, 0, H, H, res.data_ptr(), mk.data_ptr(), st)
        else:
            o.conv2d_fwd(x, w, stride=1, pad=(R - 1) // 2, out=y, stats=stats)
        torch.cuda.synchronize()
        return y
    ref = run(0)
    bad = 0
    for _ in range(reps):
        y = run(2)
        bad += int((~((y == ref) | (y.isnan() & ref.isnan()))).reshape(-1, Co).any(1).sum()) + int(y.isnan().any(-1).sum())
    lib.pfr_set_tuning(b"sconv", 1)
    total_bad += bad
    print(f"{H:3d}^2 {C:4d}->{Co:4d} R{R} stats={stats} join={join}: bad rows over {reps} launches: {bad}")
print("TOTAL BAD ROWS", total_bad)
