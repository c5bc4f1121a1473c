import copy, sys, torch
sys.path.insert(0, '.')
import pets_face_recognition_amd.models as M
torch.manual_seed(5)
bb = M.resnet18(compute_dtype=torch.float32); bb.fc = torch.nn.Linear(512, 512)
cpu = bb.train(); hip = copy.deepcopy(bb).to('cuda').train()
g = torch.Generator().manual_seed(9)
xs = [torch.rand(3, 3, 64, 64, generator=g) for _ in range(2)]
d = [torch.randn(3, 512, generator=g) for _ in range(2)]
def flat(m): return torch.cat([p.grad.flatten().cpu().double() for p in m.parameters()])
def zero(m):
    for p in m.parameters(): p.grad = None
cos = lambda a, b: torch.nn.functional.cosine_similarity(a, b, dim=0).item()
# single forward each
res = {}
for i in range(2):
    zero(cpu); cpu(xs[i]).backward(d[i]); gc = flat(cpu)
    zero(hip); hip(xs[i].cuda()).backward(d[i].cuda()); torch.cuda.synchronize(); gh = flat(hip)
    print('single', i, 'cos cpu/hip', cos(gc, gh), 'norm ratio', (gh.norm()/gc.norm()).item())
    res[i] = (gc, gh)
# sequential accumulate (fwd, bwd, fwd, bwd) without zeroing
zero(hip)
for i in range(2): hip(xs[i].cuda()).backward(d[i].cuda())
torch.cuda.synchronize(); gh_seq = flat(hip)
print('seq-acc vs sum of singles (hip)', cos(gh_seq, res[0][1] + res[1][1]), ((gh_seq - res[0][1] - res[1][1]).norm() / gh_seq.norm()).item())
# two forwards then backward
zero(hip)
e = [hip(xs[i].cuda()) for i in range(2)]
(e[0] * d[0].cuda()).sum().add((e[1] * d[1].cuda()).sum()).backward()
torch.cuda.synchronize(); gh2 = flat(hip)
print('2fwd vs sum of singles (hip)', cos(gh2, res[0][1] + res[1][1]), ((gh2 - res[0][1] - res[1][1]).norm() / gh2.norm()).item())
print('sum singles cpu vs hip', cos(res[0][0] + res[1][0], res[0][1] + res[1][1]))
names = [n for n, _ in hip.named_parameters()]
o = 0
for n, p in hip.named_parameters():
    k = p.numel(); a = gh2[o:o+k]; b = (res[0][1] + res[1][1])[o:o+k]; o += k
    c = cos(a, b)
    if c < 0.999: print('  ', n, c, a.norm().item(), b.norm().item())
