"""Which part of the streaming Linear kernel costs what: pfr_set_tuning("slin_dbg", bits) timing runs (results are wrong with bits set)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import lib, ops
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")
for M, K, N, form in [(401408, 96, 96, "res"), (401408, 96, 96, "bias"), (401408, 96, 96, "plain"), (401408, 384, 96, "plain"), (401408, 96, 288, "bias")]:
    x = torch.randn(M, 1, 1, K, device="cuda").bfloat16(); w = (torch.randn(N, 1, 1, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, 1, 1, N, device="cuda").bfloat16() if form == "res" else None
    y = torch.empty(M, 1, 1, N, device="cuda", dtype=torch.bfloat16)
    lib.pfr_set_tuning(b"slin", 2)
    out = []
    for dbg in (0, 1, 4, 5, 8):
        lib.pfr_set_tuning(b"slin_dbg", dbg)
        run = lambda: ops.conv2d_fwd(x, w, bias=None if form == "plain" else bias, residual=res, out=y)
        run(); torch.cuda.synchronize(); tt = 0.0
        for _ in range(5):
            big.add_(1.0)
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize(); tt += a.elapsed_time(b) / 5
        out.append(f"dbg{dbg} {tt*1e3:6.1f}")
    lib.pfr_set_tuning(b"slin_dbg", 0)
    print(f"M {M} K {K} N {N} {form:6s}: " + "  ".join(out) + "   (1 no stores, 4 no MFMAs, 5 both, 8 stores without the EXP_CNT wait)", flush=True)
