import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pets_face_recognition_amd._hip import lib
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for rows, C in [(128 * 3136, 96), (128 * 784, 192), (128 * 196, 384), (128 * 49, 768), (128 * 784, 384)]:
    x = torch.randn(rows, C, device=dev).bfloat16(); dy = torch.randn(rows, C, device=dev).bfloat16(); dres = torch.randn(rows, C, device=dev).bfloat16()
    y = torch.empty_like(x); dx = torch.empty_like(x)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    nb = lib.pfr_layernorm_bwd_blocks(rows)
    part = torch.empty(2, nb, C, device=dev)
    f = t(lambda: lib.pfr_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 1, rows, C, 1e-5, st))
    bw = t(lambda: lib.pfr_layernorm_bwd(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), dres.data_ptr(), dx.data_ptr(), part.data_ptr(), 1, rows, C, st))
    mb = rows * C * 2 / 1e6
    print(f"rows {rows} C {C}: fwd {f:.1f} us = {2 * mb / f / 1e6 * 1e6 / 1e6:.2f} TB/s   bwd {bw:.1f} us = {4 * mb / bw / 1e6 * 1e6 / 1e6:.2f} TB/s")
