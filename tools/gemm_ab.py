"""A/B of the conv / GEMM kernels on the launch geometries of a ResNet-50 train step (forward + data-gradient launches):
one-tile-per-workgroup kernel (igemm_p = 0) vs the persistent kernel (igemm_p = 2) with each of its tiles; outputs must be
bit-identical.  Cold timings (a 1 GiB stream between launches evicts L2 / MALL).   python tools/gemm_ab.py [out.json] [reps]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import ops, lib

OUT = sys.argv[1] if len(sys.argv) > 1 else None
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(os.environ.get("AB_BATCH", "256"))
# (H, C, Cout, R, stride, dil_log2, count)  — forward (stats) and dgrad (no stats) launches of ResNet-50 @224, bs 256
FWD = [(56, 64, 64, 1, 1, 2), (56, 64, 64, 3, 1, 3), (56, 64, 256, 1, 1, 4), (56, 256, 64, 1, 1, 2), (56, 256, 128, 1, 1, 1),
       (56, 128, 128, 3, 2, 1), (28, 128, 512, 1, 1, 4), (56, 256, 512, 1, 2, 1), (28, 512, 128, 1, 1, 3), (28, 128, 128, 3, 1, 3),
       (28, 512, 256, 1, 1, 1), (28, 256, 256, 3, 2, 1), (14, 256, 1024, 1, 1, 6), (28, 512, 1024, 1, 2, 1), (14, 1024, 256, 1, 1, 5),
       (14, 256, 256, 3, 1, 5), (14, 1024, 512, 1, 1, 1), (14, 512, 512, 3, 2, 1), (7, 512, 2048, 1, 1, 3), (14, 1024, 2048, 1, 2, 1),
       (7, 2048, 512, 1, 1, 2), (7, 512, 512, 3, 1, 2)]
cases = []
for (H, C, Co, R, s, n) in FWD:
    cases.append(("fwd", H, C, Co, R, s, 0, n))
    # data gradient: a conv over dy [OH] with Cout channels producing Cin channels at H, input dilation log2(stride)
    OH = H // s
    if not (H == 56 and C == 64 and R == 1 and Co == 64 and False):
        cases.append(("dgrad", OH, Co, C, R, 1, {1: 0, 2: 1}[s], n))

big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")
rows = []
for kind, H, C, Co, R, s, dil, n in cases:
    if kind == "dgrad" and H * (1 << dil) == 56 and Co == 3:
        continue
    pad = {1: 0, 3: 1}[R]
    x = torch.randn(B, H, H, C, device="cuda").bfloat16()
    w = (torch.randn(Co, R, R, C, device="cuda") / (C * R * R) ** 0.5).bfloat16()
    if kind == "fwd":
        OH = (H + 2 * pad - R) // s + 1
        kw = dict(stride=s, pad=pad, stats=True)
    else:
        OH = H << dil
        kw = dict(stride=1, pad=R - 1 - pad, idil_log2=dil, out_hw=(OH, OH), stats=False)
    fl = 2.0 * B * OH * OH * Co * R * R * C / (4 ** dil)
    by = (x.numel() + B * OH * OH * Co + w.numel()) * 2
    res = {}
    ref = None
    variants = [("old", 0, -1, 4, 0), ("p128x128", 2, 0, 8, 0)]   # (the wave-specialised variant was retired in round 4: profiles/r02_gemm_ab*.json)
    for name, mode, tile, kch, pf in variants:
        if pf == 2 and C % 64:
            continue
        lib.pfr_set_tuning(b"igemm_p", mode)
        lib.pfr_set_tuning(b"igemm_ptile", tile)
        lib.pfr_set_tuning(b"igemm_pkch", kch)
        lib.pfr_set_tuning(b"igemm_ppf", pf)
        y, part = ops.conv2d_fwd(x, w, **kw)
        torch.cuda.synchronize()
        if ref is None:
            ref = y.clone()
            refstat = None
            if part is not None:
                M = y.numel() // Co
                mt = ops.conv2d_fwd.last_mt
                refstat = [t.clone() for t in ops.bn_finalize(part, mt, M, None, None, 1e-5, 0.1, None, None)[:2]]
        else:
            same = torch.equal(y, ref)
            if part is not None:
                M = y.numel() // Co
                mt = ops.conv2d_fwd.last_mt
                st = ops.bn_finalize(part, mt, M, None, None, 1e-5, 0.1, None, None)[:2]
                same = same and all(torch.allclose(a, b_, rtol=2e-4, atol=1e-5) for a, b_ in zip(st, refstat))
            if not same:
                res[name + "_MISMATCH"] = float((y.float() - ref.float()).abs().max())
        t = 0.0
        for _ in range(REPS):
            big.add_(1.0)
            a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            a.record(); ops.conv2d_fwd(x, w, out=y, stats_buf=part, **kw); e.record()
            torch.cuda.synchronize(); t += a.elapsed_time(e) / REPS
        res[name] = round(t * 1e3, 1)
    ideal = max(by / 5.0e12, fl / 1.2e15) * 1e6
    best = min((v, k) for k, v in res.items() if not k.endswith("MISMATCH"))
    row = dict(kind=kind, H=H, C=C, Co=Co, R=R, s=s, dil=dil, n=n, us=res, ideal_us=round(ideal, 1), best=best[1],
               tf_old=round(fl / res["old"] / 1e6, 1), tf_best=round(fl / best[0] / 1e6, 1))
    rows.append(row)
    print(f"{kind:5s} H{H:3d} C{C:4d} Co{Co:4d} R{R} s{s} d{dil} x{n}  " + " ".join(f"{k}:{v}" for k, v in res.items()) +
          f"  ideal {ideal:.0f}  best {best[1]} ({res['old'] / best[0]:.2f}x)", flush=True)
tot_old = sum(r["us"]["old"] * r["n"] for r in rows)
tot_best = sum(min(v for k, v in r["us"].items() if not k.endswith("MISMATCH")) * r["n"] for r in rows)
tot_ideal = sum(r["ideal_us"] * r["n"] for r in rows)
print(f"per step (fwd+dgrad launches): old {tot_old / 1e3:.3f} ms  best-per-layer {tot_best / 1e3:.3f} ms  ideal {tot_ideal / 1e3:.3f} ms")
bad = [r for r in rows if any(k.endswith("MISMATCH") for k in r["us"])]
print("MISMATCHES:", len(bad))
if OUT:
    json.dump(rows, open(OUT, "w"), indent=1)
