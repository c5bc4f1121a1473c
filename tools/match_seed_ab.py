"""Seeded first chunk of the gallery match (an unfused seed of N columns, then growing fused segments) against the unfused 65 536-column first
chunk, after the counted filter epilogue made early fused segments cheaper (round 4 measured the seeded form slower: profiles/r04_match_seed_ab.txt)."""
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
for rep in range(2):
    for seed in (None, 8192, 16384, 32768):
        cosine_topk(qry, gal, K, seed_cols=seed); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); sc, idx = cosine_topk(qry, gal, K, seed_cols=seed); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"seed_cols={seed}: first-contact {min(ts)*1e3:.2f} ms (median {sorted(ts)[2]*1e3:.2f}), idx checksum {int(idx.long().sum())}", flush=True)
