"""Streaming Linear kernel (csrc/pfr_slin.hip) against the tile kernel on the Swin-T stage-1 / stage-2 shapes at batch 128:
pfr_set_tuning("slin", 0 | 2), cold operands (a 1 GiB stream between launches), plain / bias+residual / GELU forms.   python tools/slin_bench.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import lib, ops
SHAPES = [(401408, 96, 96, "res"), (401408, 96, 288, "bias"), (401408, 288, 96, "plain"), (401408, 384, 96, "res"), (401408, 96, 384, "gelu"),
          (401408, 96, 384, "gelu_bwd"), (100352, 192, 192, "res"), (100352, 192, 576, "bias"), (100352, 576, 192, "plain"),
          (100352, 768, 192, "res"), (100352, 192, 768, "gelu"), (100352, 192, 768, "gelu_bwd"), (25088, 384, 384, "res"), (25088, 384, 1152, "bias")]
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for M, K, N, form in SHAPES:
    x = torch.randn(M, 1, 1, K, device="cuda").bfloat16()
    w = (torch.randn(N, 1, 1, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, 1, 1, N, device="cuda").bfloat16() if form == "res" else None
    y = torch.empty(M, 1, 1, N, device="cuda", dtype=torch.bfloat16)
    y2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16).normal_()
    def run():
        if form == "gelu":
            lib.pfr_gemm_act(x.data_ptr(), w.data_ptr(), y.data_ptr(), 1, M, K, N, bias.data_ptr(), 2, y2.data_ptr(), st)
        elif form == "gelu_bwd":
            lib.pfr_gemm_act(x.data_ptr(), w.data_ptr(), y.data_ptr(), 1, M, K, N, 0, 3, y2.data_ptr(), st)
        else:
            ops.conv2d_fwd(x, w, bias=None if form == "plain" else bias, residual=res, out=y)
    t = {}
    for mode in (0, 2):
        lib.pfr_set_tuning(b"slin", mode)
        run(); torch.cuda.synchronize()
        tt = 0.0
        for _ in range(5):
            big.add_(1.0)
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize(); tt += a.elapsed_time(b) / 5
        t[mode] = tt * 1e3
    by = 2.0 * M * (K + N * (2 if form in ("res", "gelu", "gelu_bwd") else 1))
    print(f"M {M:7d} K {K:4d} N {N:5d} {form:9s} tile {t[0]:7.1f} us  slin {t[2]:7.1f} us  ({t[0] / t[2]:.2f}x)  slin {by / t[2] / 1e6:5.2f} TB/s of algorithmic bytes", flush=True)
lib.pfr_set_tuning(b"slin", 1)
