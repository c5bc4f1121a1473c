import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pets_face_recognition_amd._hip import ops, lib
B = int(os.environ.get("WB", "8"))
cases = [("fwd", 32, 64, 128, 1, 1, 0), ("fwd", 32, 64, 64, 3, 1, 0), ("fwd", 16, 256, 256, 3, 1, 0), ("fwd", 32, 128, 256, 1, 1, 0),
         ("fwd", 32, 64, 64, 1, 1, 0), ("dgrad", 16, 128, 128, 3, 1, 1), ("dgrad", 32, 256, 64, 1, 1, 0), ("fwd", 32, 128, 128, 3, 2, 0)]
for kind, H, C, Co, R, s, dil in cases:
    pad = {1: 0, 3: 1}[R]
    torch.manual_seed(H + C)
    x = torch.randn(B, H, H, C, device="cuda").bfloat16()
    w = (torch.randn(Co, R, R, C, device="cuda") / (C * R * R) ** 0.5).bfloat16()
    if kind == "fwd":
        kw = dict(stride=s, pad=pad, stats=True)
    else:
        OH = H << dil
        kw = dict(stride=1, pad=R - 1 - pad, idil_log2=dil, out_hw=(OH, OH), stats=False)
    res = []
    for mode in (0, 2):
        lib.pfr_set_tuning(b"igemm_ws", mode)
        y, part = ops.conv2d_fwd(x, w, **kw)
        torch.cuda.synchronize()
        st = None
        if part is not None:
            M = y.numel() // Co
            mt = ops.conv2d_fwd.last_mt
            st = ops.bn_finalize(part, mt, M, None, None, 1e-5, 0.1, None, None)[:2].clone()
        res.append((y.clone(), st, part.shape if part is not None else None))
    same = torch.equal(res[0][0], res[1][0])
    sok = True if res[0][1] is None else torch.allclose(res[0][1], res[1][1], rtol=2e-4, atol=1e-5)
    print(kind, H, C, Co, R, s, dil, "M", res[0][0].numel() // Co, "equal", same, "stats", sok, res[0][2], res[1][2],
          "maxdiff", float((res[0][0].float() - res[1][0].float()).abs().max()), flush=True)
