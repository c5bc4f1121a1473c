import sys, torch
sys.path.insert(0, '.')
from oracle import resnet_ref
import pets_face_recognition_amd.models as M
arch = sys.argv[1]; HW = int(sys.argv[2]); N = int(sys.argv[3])
torch.set_num_threads(32)
sd = resnet_ref.init_state_dict(arch, 512, seed=3)
last = ".bn3.weight" if arch == "resnet50" else ".bn2.weight"
for k in sd:
    if k.startswith("layer") and k.endswith(last): sd[k] = torch.full_like(sd[k], 0.2)
g = torch.Generator().manual_seed(17)
x = torch.rand(N, 3, HW, HW, generator=g); demb = torch.randn(N, 512, generator=g) * 0.05
names = resnet_ref.param_names(sd)
def run(quant, dt64):
    f = (lambda v: v.double()) if dt64 else (lambda v: v.clone())
    ps = {k: (f(v).requires_grad_(True) if k in names else (f(v) if v.dtype.is_floating_point else v.clone())) for k, v in sd.items()}
    e = resnet_ref.forward(ps, f(x), arch, train=True, quant=quant); e.backward(f(demb))
    return torch.cat([ps[n].grad.double().flatten() for n in names])
g64 = run(None, True); gq = run(resnet_ref.bf16_round, False)
def hip(dt):
    m = getattr(M, arch)(compute_dtype=dt); m.fc = torch.nn.Linear(m.fc.in_features, 512); m.load_state_dict(sd); m = m.cuda(); m.train()
    emb = m(x.cuda()); emb.backward(demb.cuda()); torch.cuda.synchronize()
    d = dict(m.named_parameters())
    return torch.cat([d[n].grad.double().cpu().flatten() for n in names])
gh = hip(torch.bfloat16); gf = hip(torch.float32)
cos = lambda a, b: torch.nn.functional.cosine_similarity(a, b, dim=0).item()
print('cos(hip_bf16, f64) %.4f  cos(emu_bf16_fwd_only, f64) %.4f  cos(hip_bf16, emu) %.4f  cos(hip_f32, f64) %.6f' % (cos(gh, g64), cos(gq, g64), cos(gh, gq), cos(gf, g64)))
