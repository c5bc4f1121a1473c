"""Do an HBM-bound BatchNorm kernel and an MFMA-bound weight-gradient (or tile) kernel OVERLAP when they run on two streams, or do they
time-slice?  T(A alone), T(B alone), T(A || B) for n launches each.   python tools/overlap_probe.py [n]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import ops, lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = 'cuda'
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fa, fb, na, nb):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
    if fa:
        with torch.cuda.stream(sA):
            for _ in range(na):
                fa()
    if fb:
        with torch.cuda.stream(sB):
            for _ in range(nb):
                fb()
    torch.cuda.current_stream().wait_stream(sA); torch.cuda.current_stream().wait_stream(sB)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def A_bn_act(H, C):
    x = torch.randn(256, H, H, C, device=dev).bfloat16(); a = torch.ones(C, device=dev); b = torch.zeros(C, device=dev); o = torch.empty_like(x)
    return (lambda: ops.bn_act(x, a, b, out=o)), 2 * x.numel() * 2


def B_wgrad(H, C, Co, R, splits=0):
    x = torch.randn(256, H, H, C, device=dev).bfloat16(); dy = torch.randn(256, H, H, Co, device=dev).bfloat16()
    ws = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)
    lib.pfr_set_tuning(b"wgrad_splits", splits)
    out = ops.conv2d_wgrad(x, dy, R, R, 1, R // 2, workspace=ws)
    def go():
        lib.pfr_set_tuning(b"wgrad_splits", splits)
        ops.conv2d_wgrad(x, dy, R, R, 1, R // 2, out=out, workspace=ws)
    return go, 2.0 * 256 * H * H * Co * R * R * C


def B_fwd(H, C, Co, R):
    x = torch.randn(256, H, H, C, device=dev).bfloat16(); w = (torch.randn(Co, R, R, C, device=dev) / (C * R * R) ** 0.5).bfloat16()
    y, part = ops.conv2d_fwd(x, w, stride=1, pad=R // 2, stats=True)
    return (lambda: ops.conv2d_fwd(x, w, stride=1, pad=R // 2, stats=True, out=y, stats_buf=part)), 2.0 * 256 * H * H * Co * R * R * C


CASES = [
    ("bn_act 56x56x256", A_bn_act(56, 256), "wgrad3 3x3 256 @14 (MFMA)", B_wgrad(14, 256, 256, 3)),
    ("bn_act 56x56x256", A_bn_act(56, 256), "wgrad3 3x3 256 @14, 3 splits (108 WGs)", B_wgrad(14, 256, 256, 3, 3)),
    ("bn_act 28x28x512", A_bn_act(28, 512), "wgrad3 3x3 128 @28", B_wgrad(28, 128, 128, 3)),
    ("bn_act 56x56x256", A_bn_act(56, 256), "wgrad 1x1 1024->256 @14", B_wgrad(14, 1024, 256, 1)),
    ("bn_act 56x56x256", A_bn_act(56, 256), "wgrad 1x1 64->256 @56 (swgrad, HBM)", B_wgrad(56, 64, 256, 1)),
    ("bn_act 56x56x256", A_bn_act(56, 256), "fwd tile 256x256 3x3 256 @14", B_fwd(14, 256, 256, 3)),
    ("bn_act 14x14x1024", A_bn_act(14, 1024), "wgrad3 3x3 256 @14 (MFMA)", B_wgrad(14, 256, 256, 3)),
]
for na_name, (fa, bytesA), nb_name, (fb, flB) in CASES:
    for _ in range(3):
        fa(); fb()
    ta = timed(fa, None, 1, 0); ta = timed(fa, None, n, 0) / n
    tb = timed(None, fb, 0, 1); tb = timed(None, fb, 0, n) / n
    # equal total time on both streams
    na = n; nb = max(1, round(n * ta / tb))
    tab = timed(fa, fb, na, nb)
    alone = na * ta + nb * tb
    print(f"A = {na_name}: {ta:7.1f} us ({bytesA / ta / 1e6:5.2f} TB/s) | B = {nb_name}: {tb:7.1f} us ({flB / tb / 1e6:6.0f} TF/s) | "
          f"{na} x A || {nb} x B: {tab:8.0f} us vs serial {alone:8.0f} us, ideal {max(na * ta, nb * tb):8.0f} us -> overlap efficiency "
          f"{(alone - tab) / (alone - max(na * ta, nb * tb) + 1e-9):5.2f}", flush=True)
lib.pfr_set_tuning(b"wgrad_splits", 0)
