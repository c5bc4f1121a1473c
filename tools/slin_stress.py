"""Race screen of the streaming Linear kernel (csrc/pfr_slin.hip: inline-asm loads with hand-counted waits) at the full Swin-T batch-128 sizes
and UNDER LOAD: every geometry REPS times while a second stream keeps the memory system busy, each output compared bit for bit with the
tile kernel's (the small test shapes never have a deep memory queue — DESIGN.md section 6, round 3).   python tools/slin_stress.py [reps]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import lib, ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
SHAPES = [(401408, 96, 96, "res"), (401408, 96, 288, "bias"), (401408, 288, 96, "plain"), (401408, 384, 96, "res"), (401408, 96, 384, "gelu"),
          (100352, 192, 192, "res"), (100352, 192, 576, "bias"), (100352, 768, 192, "res"), (100352 + 13, 576, 192, "plain")]
side = torch.cuda.Stream()
noise = torch.empty(192 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
st = torch.cuda.current_stream().cuda_stream
bad_total = 0
for M, K, N, form in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(M + K + N)
    x = torch.randn(M, 1, 1, K, device="cuda", generator=g).bfloat16(); w = (torch.randn(N, 1, 1, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g); res = torch.randn(M, 1, 1, N, device="cuda", generator=g).bfloat16() if form == "res" else None
    y = torch.empty(M, 1, 1, N, device="cuda", dtype=torch.bfloat16); y2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    def run():
        if form == "gelu": lib.pfr_gemm_act(x.data_ptr(), w.data_ptr(), y.data_ptr(), 1, M, K, N, bias.data_ptr(), 2, y2.data_ptr(), st)
        else: ops.conv2d_fwd(x, w, bias=None if form == "plain" else bias, residual=res, out=y)
    lib.pfr_set_tuning(b"slin", 0); run(); torch.cuda.synchronize()
    ref, ref2 = y.clone(), y2.clone()
    lib.pfr_set_tuning(b"slin", 2)
    bad = 0
    for i in range(reps):
        y.zero_(); y2.zero_()
        with torch.cuda.stream(side):
            noise.mul_(1.0000001); noise.add_(1e-9)          # ~1.5 GB of traffic racing with the launch
        run()
        torch.cuda.synchronize()
        if not torch.equal(y, ref) or (form == "gelu" and not torch.equal(y2, ref2)):
            bad += 1
    lib.pfr_set_tuning(b"slin", 1)
    bad_total += bad
    print(f"M {M} K {K} N {N} {form:5s}: {reps} repetitions under load, {bad} differ from the tile kernel", flush=True)
# GELU backward + column sums (pfr_gemm_act_colsums): output against the tile kernel's bits, partial sums against the first repetition's
# (the reduction has a fixed order: they must be bit-reproducible)
M, K, N = 401408, 96, 384
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.randn(M, K, device="cuda", generator=g).bfloat16(); w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
z = torch.randn(M, N, device="cuda", generator=g).bfloat16()
ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
lib.pfr_set_tuning(b"slin", 0)
lib.pfr_gemm_act(x.data_ptr(), w.data_ptr(), ref.data_ptr(), 1, M, K, N, 0, 3, z.data_ptr(), st); torch.cuda.synchronize()
lib.pfr_set_tuning(b"slin", 1)
parts = lib.pfr_gemm_act_colsum_parts(M, K, N, 1)
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); sums = torch.empty(parts, N, device="cuda"); first = None; bad = 0
for i in range(reps):
    y.zero_(); sums.fill_(float("nan"))
    with torch.cuda.stream(side):
        noise.mul_(1.0000001); noise.add_(1e-9)
    lib.pfr_gemm_act_colsums(x.data_ptr(), w.data_ptr(), y.data_ptr(), 1, M, K, N, z.data_ptr(), sums.data_ptr(), st)
    torch.cuda.synchronize()
    if first is None:
        first = sums.clone()
        assert torch.allclose(sums.sum(0), y.float().sum(0), rtol=1e-3, atol=0.5)
    if not torch.equal(y, ref) or not torch.equal(sums, first):
        bad += 1
bad_total += bad
print(f"M {M} K {K} N {N} gelu' + column sums ({parts} partial rows): {reps} repetitions under load, {bad} differ", flush=True)
print("TOTAL mismatching repetitions:", bad_total)
sys.exit(1 if bad_total else 0)
