import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pets_face_recognition_amd._hip import lib, ops as o
DEV = "cuda"
for case in [(8, 56, 64, 256, 1, True, True), (8, 56, 64, 256, 1, False, True), (8, 56, 64, 256, 1, True, False), (8, 56, 256, 64, 1, False, True)]:
    N, H, C, Co, sd, has_res, relu = case
    g = torch.Generator().manual_seed(H * C + Co + 7)
    x = torch.randn(N, H, H, C, generator=g).to(DEV).bfloat16()
    w = (torch.randn(Co, 1, 1, C, generator=g) / C ** 0.5).to(DEV).bfloat16()
    bias = torch.randn(Co, generator=g).to(DEV)
    OH = (H - 1) // sd + 1
    res = torch.randn(N, OH, OH, Co, generator=g).to(DEV).bfloat16() if has_res else None
    outs = []
    for mode in (0, 2):
        lib.pfr_set_tuning(b"sconv", mode)
        y, _ = o.conv2d_fwd(x, w, stride=sd, pad=0, bias=bias, residual=res, out_relu=relu)
        torch.cuda.synchronize()
        outs.append(y.clone().float())
    conv = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=sd).permute(0, 2, 3, 1)
    pre = conv + bias + (res.float() if has_res else 0)
    ref = torch.relu(pre) if relu else pre
    bound = 2.0 ** -8 * ref.abs() + 2.0 ** -8 * conv.abs() + 1e-6
    for nm, y in (("tile", outs[0]), ("sconv", outs[1])):
        d = (y - ref).abs()
        bad = d > bound
        print(case, nm, "max err", d.max().item(), "n bad", int(bad.sum()), "of", d.numel())
        if bad.any():
            idx = bad.reshape(-1, Co).nonzero()[:6]
            for r, c in idx.tolist():
                print("   row", r, "col", c, "y", y.reshape(-1, Co)[r, c].item(), "ref", ref.reshape(-1, Co)[r, c].item(), "conv", conv.reshape(-1, Co)[r, c].item())
            rows = bad.reshape(-1, Co).any(1).nonzero().flatten()
            cols = bad.reshape(-1, Co).any(0).nonzero().flatten()
            print("   bad rows", rows[:12].tolist(), "n", len(rows), "bad cols", cols[:12].tolist(), "n", len(cols))
