import sys, torch
sys.path.insert(0, '.')
from oracle import resnet_ref
import pets_face_recognition_amd.models as M
arch = sys.argv[1]; HW = int(sys.argv[2]); N = int(sys.argv[3]); dt = torch.float32 if sys.argv[4] == 'f32' else torch.bfloat16
sd = resnet_ref.init_state_dict(arch, 512, seed=3)
g = torch.Generator().manual_seed(17)
x = torch.rand(N, 3, HW, HW, generator=g); demb = torch.randn(N, 512, generator=g) * 0.05
ps = {k: (v.clone().requires_grad_(True) if k in resnet_ref.param_names(sd) else v.clone()) for k, v in sd.items()}
torch.set_num_threads(32)
emb_ref = resnet_ref.forward(ps, x, arch, train=True); emb_ref.backward(demb)
# double-precision oracle to see the conditioning of the problem itself
ps64 = {k: (v.double().clone().requires_grad_(True) if k in resnet_ref.param_names(sd) else (v.double() if v.dtype.is_floating_point else v.clone())) for k, v in sd.items()}
e64 = resnet_ref.forward(ps64, x.double(), arch, train=True); e64.backward(demb.double())
m = getattr(M, arch)(compute_dtype=dt); m.fc = torch.nn.Linear(m.fc.in_features, 512); m.load_state_dict(sd); m = m.cuda(); m.train()
emb = m(x.cuda()); emb.backward(demb.cuda()); torch.cuda.synchronize()
rel = lambda a, b: ((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-30)).item()
print('emb: hip vs f64', rel(emb, e64), ' cpu-f32 vs f64', rel(emb_ref, e64))
rows = []
for n, p in m.named_parameters():
    rows.append((rel(p.grad, ps64[n].grad), rel(ps[n].grad, ps64[n].grad), n))
rows.sort(reverse=True)
for r in rows[:8]: print('hip-vs-f64 %.3e   cpuf32-vs-f64 %.3e   %s' % r)
print('--- by ratio')
rows.sort(key=lambda r: -(r[0] / (r[1] + 1e-12)))
for r in rows[:12]: print('hip-vs-f64 %.3e   cpuf32-vs-f64 %.3e   %s' % r)
