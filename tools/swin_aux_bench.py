"""Cold per-launch time and achieved HBM rate of the non-GEMM Swin-T kernels (LayerNorm forward / backward, window attention forward /
backward) at the four stage sizes of the bs-128 224x224 step.  Bytes = the tensors each launch must read and write once.
Usage (GPU box): python tools/swin_aux_bench.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pets_face_recognition_amd._hip import lib, dtype_id

dev = "cuda:0"
did = dtype_id(torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)


def timeit(fn, n=6):
    t = 0.0
    for i in range(n + 1):
        big.add_(1.0)
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        if i:
            t += a.elapsed_time(b) / n
    return t * 1e3


B = 128
ONLY_LN = "--ln" in sys.argv
for (HW, C, heads) in [(56, 96, 3), (28, 192, 6), (14, 384, 12), (7, 768, 24)]:
    rows = B * HW * HW
    x = torch.randn(rows, C, device=dev).bfloat16(); y = torch.empty_like(x); dy = torch.randn_like(x); dres = torch.randn_like(x); dx = torch.empty_like(x)
    gamma = torch.rand(C, device=dev) + 0.5; beta = torch.zeros(C, device=dev)
    mu = torch.empty(rows, device=dev); rs = torch.empty(rows, device=dev)
    t = timeit(lambda: lib.pfr_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mu.data_ptr(), rs.data_ptr(), did, rows, C, 1e-5, st))
    tc = timeit(lambda: y.copy_(x))
    print(f"(torch copy    rows {rows:7d} C {C:4d}: {tc:7.1f} us  {rows * C * 4 / tc / 1e6:6.2f} TB/s: the 1 read : 1 write reference)")
    by = rows * C * 2 * 2 + rows * 8
    print(f"layernorm_fwd  rows {rows:7d} C {C:4d}: {t:7.1f} us  {by / t / 1e6:6.2f} TB/s")
    nb = lib.pfr_layernorm_bwd_blocks(rows)
    part = torch.empty(2, nb, C, device=dev); dsum = torch.empty(nb, C, device=dev)
    t = timeit(lambda: lib.pfr_layernorm_bwd_dxsum(dy.data_ptr(), x.data_ptr(), mu.data_ptr(), rs.data_ptr(), gamma.data_ptr(), dres.data_ptr(), dx.data_ptr(),
                                                   part.data_ptr(), dsum.data_ptr(), did, rows, C, st))
    by = rows * C * 2 * 4 + rows * 8
    print(f"layernorm_bwd  rows {rows:7d} C {C:4d}: {t:7.1f} us  {by / t / 1e6:6.2f} TB/s   (dy, x, dres in; dx out; dx column sums)")
    if ONLY_LN:
        continue
    w = 7
    qkv = torch.randn(rows, 3 * C, device=dev).bfloat16(); out = torch.empty(rows, C, device=dev, dtype=torch.bfloat16)
    pos = torch.randn(2 * w - 1, 2 * w - 1, device=dev)
    tab = torch.empty(lib.pfr_window_bias_table_floats(w), device=dev)
    lib.pfr_window_bias_table(pos.data_ptr(), tab.data_ptr(), w, 3 if HW > 7 else 0, st)
    for shift in ((0, 3) if HW > 7 else (0,)):
        lib.pfr_window_bias_table(pos.data_ptr(), tab.data_ptr(), w, shift, st)
        t = timeit(lambda: lib.pfr_window_attn_fwd(qkv.data_ptr(), tab.data_ptr(), out.data_ptr(), did, B, HW, HW, heads, 32, w, shift, 32 ** -0.5, st))
        by = rows * C * 2 * 4
        print(f"attn_fwd shift {shift} rows {rows:7d} C {C:4d}: {t:7.1f} us  {by / t / 1e6:6.2f} TB/s")
        nblk = B * (HW // w) ** 2 * heads
        dpart = torch.empty(nblk, (2 * w - 1) ** 2, device=dev); dqkv = torch.empty_like(qkv); dout = torch.randn_like(out)
        t = timeit(lambda: lib.pfr_window_attn_bwd(qkv.data_ptr(), tab.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dpart.data_ptr(), did, B, HW, HW, heads, 32, w,
                                                   shift, 32 ** -0.5, st))
        by = rows * C * 2 * 7 + dpart.numel() * 4
        print(f"attn_bwd shift {shift} rows {rows:7d} C {C:4d}: {t:7.1f} us  {by / t / 1e6:6.2f} TB/s")
