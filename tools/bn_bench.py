"""The per-channel row passes of a ResNet-50 bottleneck (bs 256, bf16), each ALONE on the chip and on cold operands (buffer sets rotated
past the 256 MB last-level cache): us per launch and TB/s of the bytes the pass must move.  What the in-step durations of the same
launches (profiles/*_step_timeline.txt) are to be read against: the difference is what running next to the weight-gradient stream costs.
  python tools/bn_bench.py [reps=20]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import ops, lib, dtype_id

kw = dict(a.split('=') for a in sys.argv[1:])
reps = int(kw.get('reps', 20))
SHAPES = [(256 * 56 * 56, 64), (256 * 56 * 56, 256), (256 * 28 * 28, 128), (256 * 28 * 28, 512), (256 * 14 * 14, 256), (256 * 14 * 14, 1024),
          (256 * 7 * 7, 512), (256 * 7 * 7, 2048)]
dev = 'cuda'
_p = lambda t: 0 if t is None else t.data_ptr()
st = lambda: torch.cuda.current_stream().cuda_stream


def timed(fn, nset):
    for i in range(nset):
        fn(i)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for i in range(reps):
        fn(i % nset)
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e3


print(f'{"rows":>8s} {"C":>5s} | {"bn_act":>16s} | {"bn_act+res":>16s} | {"bwd_reduce m3":>16s} | {"bwd_apply m3":>16s} | {"bwd_apply m3+gres":>18s}   (us, TB/s)')
for rows, C in SHAPES:
    b = rows * C * 2
    nset = max(2, int(700e6 // b) + 1)
    X = [torch.randn(rows, C, device=dev).bfloat16() for _ in range(nset)]
    D = [torch.randn(rows, C, device=dev).bfloat16() for _ in range(nset)]
    Y = [torch.empty(rows, C, device=dev, dtype=torch.bfloat16) for _ in range(nset)]
    G = [torch.empty(rows, C, device=dev, dtype=torch.bfloat16) for _ in range(nset)]
    M = [torch.randint(0, 256, (rows, C // 8), device=dev, dtype=torch.uint8) for _ in range(nset)]
    a = torch.rand(C, device=dev) + 0.5; bb = torch.randn(C, device=dev)
    mean = torch.randn(C, device=dev) * 0.1; inv = torch.rand(C, device=dev) + 0.5
    coef = torch.randn(3, C, device=dev)
    did = dtype_id(torch.bfloat16)
    nb = lib.pfr_colreduce_blocks(C, did, rows)
    part = torch.empty((nb, 2, C), dtype=torch.float32, device=dev)
    r = []
    t = timed(lambda i: lib.pfr_bn_act_mask(_p(X[i]), _p(a), _p(bb), 0, 0, 0, _p(Y[i]), _p(M[i]), did, rows, C, 1, st()), nset)
    r.append((t, (2 * b + b // 16) / t / 1e6))
    t = timed(lambda i: lib.pfr_bn_act_mask(_p(X[i]), _p(a), _p(bb), _p(D[i]), 0, 0, _p(Y[i]), _p(M[i]), did, rows, C, 1, st()), nset)
    r.append((t, (3 * b + b // 16) / t / 1e6))
    t = timed(lambda i: lib.pfr_bn_bwd_reduce(_p(D[i]), _p(M[i]), _p(X[i]), _p(mean), _p(inv), 0, 0, 3, did, rows, C, _p(part), st()), nset)
    r.append((t, (2 * b + b // 16) / t / 1e6))
    t = timed(lambda i: lib.pfr_bn_bwd_apply(_p(D[i]), _p(M[i]), _p(X[i]), _p(coef), 0, 0, 3, _p(Y[i]), 0, did, rows, C, st()), nset)
    r.append((t, (3 * b + b // 16) / t / 1e6))
    t = timed(lambda i: lib.pfr_bn_bwd_apply(_p(D[i]), _p(M[i]), _p(X[i]), _p(coef), 0, 0, 3, _p(Y[i]), _p(G[i]), did, rows, C, st()), nset)
    r.append((t, (4 * b + b // 16) / t / 1e6))
    print(f'{rows:8d} {C:5d} | ' + ' | '.join(f'{t:8.1f} {bw:6.2f}' for t, bw in r), flush=True)
    del X, D, Y, G, M
    torch.cuda.empty_cache()
