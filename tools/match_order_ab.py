"""A/B of the filter GEMM's tile order inside one process (pfr_set_tuning("match_order", 0 | 1)) at the BASELINE config-5 size:
identical results (index checksum) and the time of the whole match.   python tools/match_order_ab.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import lib
from pets_face_recognition_amd.match import cosine_topk, prepare_gallery
Q, G, D, K = 10000, 1000000, 512, 100
g = torch.Generator(device="cuda").manual_seed(123)
ncls = G // 10
centers = torch.randn(ncls, D, device="cuda", generator=g)
gcls = torch.arange(ncls, device="cuda").repeat_interleave(10)[torch.randperm(G, device="cuda", generator=g)]
gal = centers[gcls] + 3.2 * torch.randn(G, D, device="cuda", generator=g)
qry = centers[torch.randint(0, ncls, (Q,), device="cuda", generator=g)] + 3.2 * torch.randn(Q, D, device="cuda", generator=g)
for rep in range(3):
    for mode in (0, 1):
        lib.pfr_set_tuning(b"match_order", mode)
        cosine_topk(qry, gal, K); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); sc, idx = cosine_topk(qry, gal, K); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        pg = prepare_gallery(gal); cosine_topk(qry, pg, K); torch.cuda.synchronize()
        tp = []
        for _ in range(5):
            t0 = time.perf_counter(); cosine_topk(qry, pg, K); torch.cuda.synchronize(); tp.append(time.perf_counter() - t0)
        del pg
        print(f"match_order={mode}: first-contact {min(ts)*1e3:.2f} ms (median {sorted(ts)[2]*1e3:.2f}), prepared gallery {min(tp)*1e3:.2f} ms, idx checksum {int(idx.long().sum())}", flush=True)
