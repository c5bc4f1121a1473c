"""How far apart are two CORRECT bf16 evaluations of the same train step?  The gradient-rounding oracle (oracle/resnet_ref.py, quant =
bf16_round_fb) is evaluated twice — fp32 arithmetic and fp64 arithmetic between the SAME bf16 storage points; the two differ only where a
pre-rounding value sits within ~1e-7 of a bf16 rounding boundary.  CPU only.   python tools/diag_bf16cond.py <arch> <HW> <N> [damp]"""
import sys, torch
sys.path.insert(0, '.')
from oracle import resnet_ref
arch = sys.argv[1]; HW = int(sys.argv[2]); N = int(sys.argv[3]); damp = len(sys.argv) > 4 and sys.argv[4] == "damp"
torch.set_num_threads(8)
sd = resnet_ref.init_state_dict(arch, 512, seed=3)
if damp:
    last = ".bn3.weight" if resnet_ref.ARCH[arch][0] == "bottleneck" else ".bn2.weight"
    for k in sd:
        if k.startswith("layer") and k.endswith(last): sd[k] = torch.full_like(sd[k], 0.2)
g = torch.Generator().manual_seed(17)
x = torch.rand(N, 3, HW, HW, generator=g); demb = torch.randn(N, 512, generator=g) * 0.05
names = resnet_ref.param_names(sd)
def run(quant, dt64):
    f = (lambda v: v.double()) if dt64 else (lambda v: v.clone())
    ps = {k: (f(v).requires_grad_(True) if k in names else (f(v) if v.dtype.is_floating_point else v.clone())) for k, v in sd.items()}
    e = resnet_ref.forward(ps, f(x), arch, train=True, quant=quant); e.backward(f(demb))
    return e.detach().double(), {n: ps[n].grad.double().flatten() for n in names}
e32, g32 = run(resnet_ref.bf16_round_fb, False)
e64, g64 = run(resnet_ref.bf16_round_fb, True)
a = torch.cat([g32[n] for n in names]); b = torch.cat([g64[n] for n in names])
cosn = {n: (g32[n] @ g64[n] / (g32[n].norm() * g64[n].norm() + 1e-30)).item() for n in names}
wk = min(cosn, key=cosn.get)
print(f"{arch} {HW} N{N} damp={damp}: two correct bf16 evaluations (fp32 vs fp64 arithmetic, same storage points): emb rel {((e32 - e64).norm() / e64.norm()).item():.3e}, "
      f"gradient rel err {((a - b).norm() / b.norm()).item():.4f}, worst per-tensor cosine {cosn[wk]:.4f} ({wk})")
