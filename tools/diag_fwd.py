import sys, torch
sys.path.insert(0, '.')
from oracle import resnet_ref
import pets_face_recognition_amd.models as M
arch = sys.argv[1]; HW = int(sys.argv[2]); N = int(sys.argv[3]); dt = torch.float32 if sys.argv[4] == 'f32' else torch.bfloat16
sd = resnet_ref.init_state_dict(arch, 512, seed=3)
g = torch.Generator().manual_seed(17)
x = torch.rand(N, 3, HW, HW, generator=g)
torch.set_num_threads(32)
taps = {}
emb_ref = resnet_ref.forward(sd, x, arch, train=True, taps=taps, quant=(resnet_ref.bf16_round if (len(sys.argv) > 5 and sys.argv[5] == 'q') else None))
m = getattr(M, arch)(compute_dtype=dt); m.fc = torch.nn.Linear(m.fc.in_features, 512); m.load_state_dict(sd); m = m.cuda(); m.train()
eng = m.hip_engine()
# grab block outputs: rebuild plan and keep references
import pets_face_recognition_amd.models._fe_engine as E
orig = eng.build_plan
outs = []
def patched(N_, H_, W_, train, wb):
    p = orig(N_, H_, W_, train, wb)
    return p
emb = m(x.cuda()); torch.cuda.synchronize()
rel = lambda a, b: ((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-30)).item()
print('emb', rel(emb, emb_ref))
plan = eng._last_plan
# block outputs are the bufs whose shape matches taps in order: find by matching shapes sequentially
names = list(taps.keys())
bufs = list(plan.bufs.values())
bi = 0
for nme in names:
    t = taps[nme].permute(0, 2, 3, 1)
    best = None
    for j in range(bi, len(bufs)):
        b = bufs[j]
        if tuple(b.shape) == tuple(t.shape) and b.dtype == dt:
            r = rel(b.float(), t)
            if best is None or r < best[0]:
                best = (r, j)
            if r < 0.5:
                break
    print(nme, 'rel err %.4e' % best[0], 'buf', best[1]); bi = best[1] + 1
