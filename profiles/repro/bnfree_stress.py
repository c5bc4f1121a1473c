"""Stress of the round-4 streaming epilogues at the bs-256 geometries (the class of defect found in round 3 — store data registers read
late under a deep vector-memory queue — only showed inside the real step): every launch is repeated with unrelated HBM traffic on a second
stream and compared BIT FOR BIT with the first result.   python profiles/repro/bnfree_stress.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pets_face_recognition_amd._hip import lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
noise = torch.empty(512 * 1024 * 1024, dtype=torch.bfloat16, device=dev)
side = torch.cuda.Stream()
lib.pfr_set_tuning(b"bnb", 2)
bad = 0
for (N, H, C, Co) in ((256, 56, 64, 256), (256, 28, 128, 512)):
    M = N * H * H
    z = torch.relu(torch.randn(M, C, device=dev, generator=g)).bfloat16()
    w = (torch.randn(Co, C, device=dev, generator=g) / C ** 0.5).bfloat16()
    res = torch.randn(M, Co, device=dev, generator=g).bfloat16()
    a1 = 1 + 0.1 * torch.randn(Co, device=dev, generator=g); b1 = 0.1 * torch.randn(Co, device=dev, generator=g)
    a2 = 1 + 0.1 * torch.randn(Co, device=dev, generator=g); b2 = 0.1 * torch.randn(Co, device=dev, generator=g)
    G_ = (torch.randn(M, Co, device=dev, generator=g) * (torch.rand(M, Co, device=dev, generator=g) > 0.5)).bfloat16()
    wcat = (torch.randn(C, Co + C, device=dev, generator=g) / (Co + C) ** 0.5).bfloat16()
    bias = 0.1 * torch.randn(C, device=dev, generator=g)
    bnx = torch.randn(M, C, device=dev, generator=g).bfloat16()
    coef = torch.stack([0.1 * torch.randn(C, device=dev, generator=g), 1 + 0.1 * torch.rand(C, device=dev, generator=g),
                        1 + 0.1 * torch.rand(C, device=dev, generator=g), 0.2 * torch.randn(C, device=dev, generator=g)])
    npart = lib.pfr_conv1x1_dgrad2_bn_parts(1, N, H, H, Co, C, C)
    ws = torch.empty(lib.pfr_gram_ws_floats(M, C), dtype=torch.float32, device=dev)
    ref = None
    for it in range(reps):
        with torch.cuda.stream(side):
            noise.add_(1)                      # unrelated traffic keeps the memory queues deep
        st = torch.cuda.current_stream().cuda_stream
        out = torch.empty(M, Co, dtype=torch.bfloat16, device=dev); mask = torch.empty(M, Co // 8, dtype=torch.uint8, device=dev)
        lib.pfr_conv1x1_bn_tail(z.data_ptr(), w.data_ptr(), out.data_ptr(), mask.data_ptr(), 1, N, H, H, C, Co, a1.data_ptr(), b1.data_ptr(),
                                res.data_ptr(), 0, 0, st)
        out2 = torch.empty_like(out); mask2 = torch.empty_like(mask)
        lib.pfr_conv1x1_bn_tail(z.data_ptr(), w.data_ptr(), out2.data_ptr(), mask2.data_ptr(), 1, N, H, H, C, Co, a1.data_ptr(), b1.data_ptr(),
                                res.data_ptr(), a2.data_ptr(), b2.data_ptr(), st)
        dx = torch.empty(M, C, dtype=torch.bfloat16, device=dev); part = torch.empty(npart, 2, C, dtype=torch.float32, device=dev)
        lib.pfr_conv1x1_dgrad2_bn(G_.data_ptr(), z.data_ptr(), wcat.data_ptr(), bias.data_ptr(), dx.data_ptr(), 1, N, H, H, Co, C, C,
                                  bnx.data_ptr(), coef.data_ptr(), part.data_ptr(), st)
        gram = torch.empty(C * C + C, dtype=torch.float32, device=dev)
        lib.pfr_gram_colsum(z.data_ptr(), 1, M, C, gram.data_ptr(), ws.data_ptr(), st)
        torch.cuda.synchronize()
        cur = (out, mask, out2, mask2, dx, part, gram)
        if ref is None:
            ref = [t.clone() for t in cur]
            assert all(torch.isfinite(t.float()).all() for t in (out, out2, dx, part, gram))
        else:
            for name, a, b in zip(("tail", "tail mask", "tail(proj)", "tail(proj) mask", "dgrad2", "dgrad2 sums", "gram"), ref, cur):
                if not torch.equal(a, b):
                    bad += 1
                    print(f"MISMATCH {name} at {H}x{H} it {it}: {(a.float() - b.float()).abs().max().item()}")
    print(f"{H}x{H} C{C}->{Co}: {reps} repetitions under load, bit-identical" if not bad else f"{H}x{H}: {bad} mismatches")
lib.pfr_set_tuning(b"bnb", 0)
print("OK" if not bad else "FAILED")
