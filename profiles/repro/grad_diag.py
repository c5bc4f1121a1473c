"""Which kernel family makes the bs-256 ResNet-50 gradients differ / non-finite: compares every parameter gradient of one train
step between kernel selections (same inputs, same weights).  usage: python profiles/repro/grad_diag.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pets_face_recognition_amd.models as M
from pets_face_recognition_amd._hip import lib
dev = "cuda:0"
torch.manual_seed(0)
m = M.resnet50(compute_dtype=torch.bfloat16)
m.fc = torch.nn.Linear(2048, 512)
m = m.to(dev).train()
g = torch.Generator().manual_seed(1)
x = torch.rand(256, 3, 224, 224, generator=g).to(dev)
dy = torch.randn(256, 512, generator=g).to(dev)

def grads(sconv, sconv3, il):
    os.environ["PFR_SCONV_INTERLEAVE"] = il
    lib.pfr_set_tuning(b"sconv", sconv); lib.pfr_set_tuning(b"sconv3", sconv3)
    eng = m.hip_engine(dev); eng.plans.clear()
    for p in m.parameters(): p.grad = None
    out = m(x)
    out.backward(dy)
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters()}, out.detach().clone()

ref, oref = grads(0, 0, "1")
for name, cfg in (("sconv only", (1, 0)),):
    for rep in range(1):
        g2, o2 = grads(cfg[0], cfg[1], "1")
        bad = [(n, bool(torch.isfinite(v).all()), float((v - ref[n]).abs().max())) for n, v in g2.items() if not torch.equal(v, ref[n])]
        print(name, "rep", rep, "emb equal", torch.equal(o2, oref), "params differing", len(bad), "nonfinite", [b[0] for b in bad if not b[1]][:6], bad[:3])

print("forward: rel diff of embeddings", float((o2 - oref).norm() / oref.norm()), "finite", bool(torch.isfinite(o2).all()))
names = [n for n, _ in m.named_parameters()]
fin = [bool(torch.isfinite(g2[n]).all()) for n in names]
last_bad = max([i for i, f in enumerate(fin) if not f] or [0])
print("params", len(names), "non-finite", fin.count(False), "last non-finite (forward order):", names[last_bad], "next finite:", names[last_bad + 1:last_bad + 4])
for n in names[last_bad - 2:last_bad + 6]:
    print(" ", n, "finite", bool(torch.isfinite(g2[n]).all()), "maxdiff", float((g2[n] - ref[n]).abs().max()), "refmax", float(ref[n].abs().max()))
