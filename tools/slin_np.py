import sys, os, torch
sys.path.insert(0, '/root/repo')
from pets_face_recognition_amd._hip import lib, ops
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for M, K, N, form in [(100352, 192, 192, "res"), (100352, 192, 576, "bias"), (100352, 192, 768, "gelu"), (401408, 96, 384, "gelu"), (401408, 96, 384, "bias"), (401408, 384, 96, "res")]:
    x = torch.randn(M, 1, 1, K, device="cuda").bfloat16(); w = (torch.randn(N, 1, 1, K, device="cuda") / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, 1, 1, N, device="cuda").bfloat16() if form == "res" else None
    y = torch.empty(M, 1, 1, N, device="cuda", dtype=torch.bfloat16); y2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    def run():
        if form == "gelu": lib.pfr_gemm_act(x.data_ptr(), w.data_ptr(), y.data_ptr(), 1, M, K, N, bias.data_ptr(), 2, y2.data_ptr(), st)
        else: ops.conv2d_fwd(x, w, bias=bias, residual=res, out=y)
    out = []
    for mode, npf in ((0, 0), (2, 192), (2, 96), (2, 64)):
        lib.pfr_set_tuning(b"slin", mode); lib.pfr_set_tuning(b"slin_np", npf)
        run(); torch.cuda.synchronize(); tw = 0.0
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): run()
        b.record(); torch.cuda.synchronize(); tw = a.elapsed_time(b) / 10
        out.append(f"{'tile' if mode == 0 else 'np' + str(npf)} {tw*1e3:6.1f}")
    lib.pfr_set_tuning(b"slin", 1); lib.pfr_set_tuning(b"slin_np", 0)
    print(f"M {M} K {K} N {N} {form:5s} (warm, 10 back-to-back): " + "  ".join(out), flush=True)
