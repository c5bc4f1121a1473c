"""The GELU GEMMs of Swin-T stages 3-4 (fc1 forward: act 2, writes the pre-activation and its GELU; fc2 data gradient: act 3, GELU backward)
under every forced tile of the tile kernel and the streaming kernel (slin = 2), cold operands.   python tools/gemm_act_tiles.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import lib, ops
TILES = {-1: "heur", 0: "128x128", 1: "64x128", 4: "256x256", 5: "256x128"}
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for M, K, N in [(25088, 384, 1536), (6272, 768, 3072), (100352, 192, 768)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    y2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16).normal_()
    for form, act, bp in (("bias", 0, bias.data_ptr()), ("gelu", 2, bias.data_ptr()), ("gelu_bwd", 3, 0)):
        if act == 0:
            x4, w4, y4 = x.view(M, 1, 1, K), w.view(N, 1, 1, K), y.view(M, 1, 1, N)
            run = lambda: ops.conv2d_fwd(x4, w4, bias=bias, out=y4)
        else:
            run = lambda: lib.pfr_gemm_act(x.data_ptr(), w.data_ptr(), y.data_ptr(), 1, M, K, N, bp, act, y2.data_ptr(), st)
        out = []
        for slin, tile in [(0, t) for t in TILES] + [(2, -1)]:
            lib.pfr_set_tuning(b"slin", slin); lib.pfr_set_tuning(b"igemm_tile", tile)
            run(); torch.cuda.synchronize()
            tt = 0.0
            for _ in range(5):
                big.add_(1.0)
                a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
                a.record(); run(); b.record(); torch.cuda.synchronize(); tt += a.elapsed_time(b) / 5
            out.append(f"{'slin' if slin else TILES[tile]}: {tt * 1e3:6.1f}")
        fl = 2.0 * M * K * N
        print(f"M {M:6d} K {K:4d} N {N:4d} {form:9s} " + "  ".join(out) + f"   (MFMA at 1.06 PF: {fl / 1.06e15 * 1e6:.1f} us, bytes at 6.3 TB/s: {2.0 * (M * K + 2 * M * N) / 6.3e12 * 1e6:.1f} us)", flush=True)
lib.pfr_set_tuning(b"slin", 1); lib.pfr_set_tuning(b"igemm_tile", -1)
