"""Schedule parameters of the gallery match at 10 k x 1 M x 512 (bf16 candidates + fp32 re-score): seed columns, merge interval, chunk, slack.
   python tools/match_sched_ab.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd.match import cosine_topk
Q, G, D, K = 10000, 1000000, 512, 100
g = torch.Generator(device="cuda").manual_seed(123)
ncls = G // 10
centers = torch.randn(ncls, D, device="cuda", generator=g)
gcls = torch.arange(ncls, device="cuda").repeat_interleave(10)[torch.randperm(G, device="cuda", generator=g)]
gal = centers[gcls] + 3.2 * torch.randn(G, D, device="cuda", generator=g)
qry = centers[torch.randint(0, ncls, (Q,), device="cuda", generator=g)] + 3.2 * torch.randn(Q, D, device="cuda", generator=g)
CFG = [dict(), dict(seed_cols=16384, merge_every=3), dict(chunk=131072, seed_cols=32768, merge_every=3), dict(chunk=131072, seed_cols=16384),
       dict(chunk=131072, seed_cols=16384, merge_every=3), dict(seed_cols=16384, merge_every=4), dict(chunk=262144, seed_cols=32768), dict(slack=150, certify=False)]
for rep in range(2):
    for kw in CFG:
        cosine_topk(qry, gal, K, **kw); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); sc, idx = cosine_topk(qry, gal, K, **kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        from pets_face_recognition_amd import match as _m
        print(f"{str(kw):60s} {_m.last_match_stats.get('widened')} {_m.last_match_stats.get('max_selection_error', 0):.2e} min {min(ts)*1e3:6.2f} ms  median {sorted(ts)[2]*1e3:6.2f}  idx checksum {int(idx.long().sum())}", flush=True)
