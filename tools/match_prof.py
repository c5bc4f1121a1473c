import sys, os, torch, time
sys.path.insert(0, '/root/repo')
from pets_face_recognition_amd.match import cosine_topk
from pets_face_recognition_amd._hip import set_tracer, EventTracer
Q, G, D, K = 10000, 1000000, 512, 100
g = torch.Generator(device='cuda').manual_seed(1)
gal = torch.randn(G, D, device='cuda', generator=g); qry = torch.randn(Q, D, device='cuda', generator=g)
cosine_topk(qry, gal, K, compute_dtype=torch.bfloat16); torch.cuda.synchronize()
tr = EventTracer(); set_tracer(tr)
t0 = time.perf_counter(); cosine_topk(qry, gal, K, compute_dtype=torch.bfloat16); torch.cuda.synchronize(); print("traced total", time.perf_counter() - t0)
set_tracer(None)
for k, (n, ms) in sorted(tr.summary().items(), key=lambda kv: -kv[1][1]): print(k, n, round(ms, 2), "ms")
