"""HIP kernel numerics vs plain PyTorch fp32 CPU references of the same op (called through the C-ABI)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def ops():
    from pets_face_recognition_amd._hip import ops as o
    return o


def tol(dtype):
    return dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=2e-4, atol=2e-4)


def rel_err(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


def nhwc(x):  # NCHW -> NHWC contiguous
    return x.permute(0, 2, 3, 1).contiguous()


CONV_CASES = [
    # N, H, W, C, Cout, R, stride, pad
    (2, 16, 16, 64, 64, 1, 1, 0),
    (2, 16, 16, 64, 256, 1, 1, 0),
    (3, 14, 14, 128, 128, 3, 1, 1),
    (2, 15, 15, 64, 128, 3, 2, 1),
    (2, 16, 16, 256, 512, 1, 2, 0),
    (2, 32, 32, 8, 64, 7, 2, 3),
    (5, 7, 7, 512, 512, 3, 1, 1),
    (1, 9, 11, 32, 96, 3, 1, 1),
    (2, 16, 16, 64, 128, 3, 2, 1),     # stride-2 3x3 with even extent: parity-class data gradient
    (3, 28, 28, 128, 128, 3, 2, 1),
    (64, 28, 28, 128, 256, 3, 1, 1),   # 8-wave 256x256 / 256x128 tiles (bf16)
    (180, 14, 14, 256, 128, 3, 1, 1),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case, dtype):
    o = ops()
    N, H, W, C, Cout, R, stride, pad = case
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Cout, C, R, R, generator=g) / math.sqrt(C * R * R)
    if dtype == torch.bfloat16:  # compare on identical (bf16-representable) inputs
        x = x.bfloat16().float()
        w = w.bfloat16().float()
    x.requires_grad_(True)
    w.requires_grad_(True)
    y = F.conv2d(x, w, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    y.backward(dy)

    xd = nhwc(x.detach()).to(DEV, dtype)
    wd = w.detach().permute(0, 2, 3, 1).contiguous().to(DEV, dtype)  # [Cout,R,S,C]
    yd, part = o.conv2d_fwd(xd, wd, stride=stride, pad=pad, stats=True)
    torch.cuda.synchronize()
    yr = nhwc(y.detach())
    assert yd.shape == yr.shape
    e = rel_err(yd.cpu(), yr)
    assert e < (1e-2 if dtype == torch.bfloat16 else 1e-5), f"fwd rel err {e}"
    # fused per-channel statistics (per-m-tile mean / M2, merged by pfr_bn_finalize) of the stored output
    from pets_face_recognition_amd._hip import lib
    M = yd.numel() // Cout
    did = 0 if dtype == torch.float32 else 1
    part_mt = lib.pfr_conv2d_mtile(N, H, W, C, Cout, R, R, stride, pad, yd.shape[1], yd.shape[2], did, did, 0)
    coef = o.bn_finalize(part, part_mt, M, None, None, 1e-5, 0.1, None, None)
    torch.cuda.synchronize()
    ydf = yd.double().cpu().reshape(-1, Cout)
    assert torch.allclose(coef[0].cpu().double(), ydf.mean(0), rtol=1e-4, atol=1e-5)
    assert torch.allclose(coef[1].cpu().double(), 1.0 / torch.sqrt(ydf.var(0, unbiased=False) + 1e-5), rtol=1e-4, atol=1e-5)

    dyd = nhwc(dy).to(DEV, dtype)
    wt = o.weight_dgrad_layout(wd)
    dxd = o.conv2d_dgrad(dyd, wt, (H, W), stride, pad, R, R)
    torch.cuda.synchronize()
    e = rel_err(dxd.cpu(), nhwc(x.grad))
    assert e < (1e-2 if dtype == torch.bfloat16 else 1e-5), f"dgrad rel err {e}"

    dwd = o.conv2d_wgrad(xd, dyd, R, R, stride, pad)
    torch.cuda.synchronize()
    e = rel_err(dwd.cpu(), w.grad.permute(0, 2, 3, 1))
    assert e < (1e-2 if dtype == torch.bfloat16 else 1e-5), f"wgrad rel err {e}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_prologue_bias_accumulate(dtype):
    o = ops()
    g = torch.Generator().manual_seed(7)
    N, H, W, C, Cout = 2, 10, 10, 64, 72
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Cout, C, 3, 3, generator=g) / 24
    sc = torch.rand(C, generator=g) + 0.5
    sh = torch.randn(C, generator=g) * 0.3
    bias = torch.randn(Cout, generator=g)
    y0 = torch.randn(N, Cout, H, W, generator=g)
    if dtype == torch.bfloat16:
        x, w, y0 = x.bfloat16().float(), w.bfloat16().float(), y0.bfloat16().float()
    xa = F.relu(x * sc[None, :, None, None] + sh[None, :, None, None])
    if dtype == torch.bfloat16:
        xa = xa.bfloat16().float()
    ref = F.relu(F.conv2d(xa, w, bias=bias, padding=1) + y0)
    xd = nhwc(x).to(DEV, dtype)
    wd = w.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    out = nhwc(y0).to(DEV, dtype)
    o.conv2d_fwd(xd, wd, stride=1, pad=1, out=out, bias=bias.to(DEV), accumulate=True, out_relu=True,
                 pro=(sc.to(DEV), sh.to(DEV), 1))
    torch.cuda.synchronize()
    e = rel_err(out.cpu(), nhwc(ref))
    assert e < (1e-2 if dtype == torch.bfloat16 else 1e-5), e
    # wgrad with the same prologue
    dy = torch.randn(N, Cout, H, W, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    wq = w.clone().requires_grad_(True)
    F.conv2d(xa, wq, padding=1).backward(dy)
    dwd = o.conv2d_wgrad(xd, nhwc(dy).to(DEV, dtype), 3, 3, 1, 1, pro=(sc.to(DEV), sh.to(DEV), 1))
    torch.cuda.synchronize()
    e = rel_err(dwd.cpu(), wq.grad.permute(0, 2, 3, 1))
    assert e < (1e-2 if dtype == torch.bfloat16 else 1e-5), e


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_f32_out_odd_cols(dtype):
    """plain GEMM (1x1 conv, H=W=1) with fp32 output, N not a multiple of the tile and a padded row pitch"""
    o = ops()
    g = torch.Generator().manual_seed(3)
    B, D, C = 37, 512, 100
    x = torch.randn(B, D, generator=g)
    w = torch.randn(C, D, generator=g) / 22
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = x @ w.t()
    ld = 104
    out = torch.zeros(B, 1, 1, ld, dtype=torch.float32, device=DEV)
    xd = x.view(B, 1, 1, D).to(DEV, dtype)
    wd = w.view(C, 1, 1, D).to(DEV, dtype)
    from pets_face_recognition_amd._hip import lib, dtype_id
    lib.pfr_conv2d_fwd(xd.data_ptr(), wd.data_ptr(), out.data_ptr(), dtype_id(dtype), 0, B, 1, 1, D, C, 1, 1, 1, 0, 0, 1, 1,
                       ld, 0, 0, 0, 0, 0, 0, 0, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.view(B, ld).cpu()
    assert rel_err(got[:, :C], ref) < (1e-2 if dtype == torch.bfloat16 else 1e-5)
    assert (got[:, C:] == 0).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [64, 256, 2048])
def test_batchnorm_train_fwd_bwd(dtype, C):
    o = ops()
    g = torch.Generator().manual_seed(11)
    N, H, W = 4, 9, 9
    x = torch.randn(N, C, H, W, generator=g) * 2 + 0.5
    res = torch.randn(N, C, H, W, generator=g)
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.2
    if dtype == torch.bfloat16:
        x, res = x.bfloat16().float(), res.bfloat16().float()
    x.requires_grad_(True)
    res.requires_grad_(True)
    gam = gamma.clone().requires_grad_(True)
    bet = beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    y = F.relu(F.batch_norm(x, rm, rv, gam, bet, training=True, momentum=0.1, eps=1e-5) + res)
    dy = torch.randn(y.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    y.backward(dy)

    xd, resd = nhwc(x.detach()).to(DEV, dtype), nhwc(res.detach()).to(DEV, dtype)
    part, rpp = o.bn_stats(xd)
    rmd, rvd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    coef = o.bn_finalize(part, rpp, N * H * W, gamma.to(DEV), beta.to(DEV), 1e-5, 0.1, rmd, rvd)
    yd = o.bn_act(xd, coef[2], coef[3], x2=resd, relu=True)
    torch.cuda.synchronize()
    t = tol(dtype)
    assert torch.allclose(rmd.cpu(), rm, rtol=1e-4, atol=1e-5)
    assert torch.allclose(rvd.cpu(), rv, rtol=1e-4, atol=1e-5)
    assert torch.allclose(yd.float().cpu(), nhwc(y.detach()), **t)
    dx, gres, dgam, dbet = o.bn_bwd(nhwc(dy).to(DEV, dtype), xd, coef[0], coef[1], gamma.to(DEV), N * H * W, mask_mode=1,
                                     out_act=yd, want_gres=True)
    torch.cuda.synchronize()
    # compare against autograd on values where the ReLU mask agrees (bf16 output rounding can flip exact zeros)
    assert rel_err(dx.cpu(), nhwc(x.grad)) < (2e-2 if dtype == torch.bfloat16 else 1e-4)
    assert rel_err(gres.cpu(), nhwc(res.grad)) < (2e-2 if dtype == torch.bfloat16 else 1e-5)
    assert rel_err(dgam.cpu(), gam.grad) < (2e-2 if dtype == torch.bfloat16 else 1e-4)
    assert rel_err(dbet.cpu(), bet.grad) < (2e-2 if dtype == torch.bfloat16 else 1e-4)
    # the 1-bit ReLU mask written by the forward pass (mask_mode 3) gives bit-identical gradients to re-reading the output
    yd2, mask = o.bn_act(xd, coef[2], coef[3], x2=resd, relu=True, want_mask=True)
    dx3, gres3, dgam3, dbet3 = o.bn_bwd(nhwc(dy).to(DEV, dtype), xd, coef[0], coef[1], gamma.to(DEV), N * H * W, mask_mode=3,
                                         out_act=mask, want_gres=True)
    torch.cuda.synchronize()
    assert torch.equal(yd2, yd)
    if dtype == torch.float32:   # (a bf16 output can round a tiny positive pre-activation to +0: mask bit set, out > 0 false)
        assert torch.equal(dx3, dx) and torch.equal(gres3, gres) and torch.equal(dgam3, dgam) and torch.equal(dbet3, dbet)
    else:
        assert rel_err(dx3.cpu(), dx.cpu()) < 1e-2 and rel_err(dgam3.cpu(), dgam.cpu()) < 1e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_relu_maxpool_and_avgpool(dtype):
    o = ops()
    g = torch.Generator().manual_seed(5)
    N, C, H, W = 3, 64, 18, 18
    x = torch.randn(N, C, H, W, generator=g)
    sc = torch.rand(C, generator=g) + 0.5
    sh = torch.randn(C, generator=g) * 0.5
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    z = F.relu(x * sc[None, :, None, None] + sh[None, :, None, None])
    if dtype == torch.bfloat16:
        z = z.bfloat16().float()
    z.requires_grad_(True)
    y = F.max_pool2d(z, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    y.backward(dy)
    xd = nhwc(x).to(DEV, dtype)
    yd, idx = o.bn_relu_maxpool_fwd(xd, sc.to(DEV), sh.to(DEV))
    dz = o.maxpool_bwd(nhwc(dy).to(DEV, dtype), idx, (H, W))
    torch.cuda.synchronize()
    assert torch.allclose(yd.float().cpu(), nhwc(y.detach()), **tol(dtype))
    assert rel_err(dz.cpu(), nhwc(z.grad)) < (1e-2 if dtype == torch.bfloat16 else 1e-6)
    # global average pool
    a = torch.randn(N, 7, 7, 256, generator=g)
    ad = a.to(DEV, dtype)
    p = o.avgpool_fwd(ad)
    torch.cuda.synchronize()
    assert torch.allclose(p.float().cpu(), ad.float().cpu().mean(dim=(1, 2)), **tol(dtype))
    d = o.avgpool_bwd(p, (7, 7))
    torch.cuda.synchronize()
    assert torch.allclose(d.float().cpu(), (p.float().cpu() / 49)[:, None, None, :].expand(N, 7, 7, 256), **tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C,H,W", [(2, 64, 17, 17), (2, 64, 18, 22), (1, 64, 112, 112), (2, 16, 6, 4)])
def test_maxpool_backward_even_and_odd_sizes(dtype, N, C, H, W):
    """pfr_maxpool_bwd against torch on the sizes of both of its kernels: even H, W take the 2 x 2 input-block gather (112 x 112 is the
    stem's), odd ones the per-pixel gather (reference: nn.MaxPool2d(3, 2, 1) of torchvision's ResNet stem, models/__init__.py backbones)"""
    o = ops()
    g = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(N, C, H, W, generator=g)
    sc = torch.rand(C, generator=g) + 0.5
    sh = torch.randn(C, generator=g) * 0.5
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    z = F.relu(x * sc[None, :, None, None] + sh[None, :, None, None])
    if dtype == torch.bfloat16:
        z = z.bfloat16().float()
    z.requires_grad_(True)
    y = F.max_pool2d(z, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    y.backward(dy)
    yd, idx = o.bn_relu_maxpool_fwd(nhwc(x).to(DEV, dtype), sc.to(DEV), sh.to(DEV))
    dz = o.maxpool_bwd(nhwc(dy).to(DEV, dtype), idx, (H, W))
    torch.cuda.synchronize()
    assert rel_err(dz.cpu(), nhwc(z.grad)) < (1e-2 if dtype == torch.bfloat16 else 1e-6)


def _arcface_ref(x, w, label, s, m, easy, cosface=False, gamma=0.0):
    cos = F.linear(F.normalize(x), F.normalize(w))
    if cosface:
        phi = cos - m
    else:
        sine = torch.sqrt(1.0 - cos * cos)
        phi = cos * math.cos(m) - sine * math.sin(m)
        if easy:
            phi = torch.where(cos > 0, phi, cos)
        else:
            phi = torch.where(cos > math.cos(math.pi - m), phi, cos - math.sin(math.pi - m) * m)
    oh = torch.zeros_like(cos)
    oh.scatter_(1, label.view(-1, 1), 1)
    logits = s * (oh * phi + (1 - oh) * cos)
    logp = F.cross_entropy(logits, label, reduction="none")
    p = torch.exp(-logp)
    return logits, ((1 - p) ** gamma * logp).mean(), cos


@pytest.mark.parametrize("mode,gamma", [("arc", 0.0), ("arc_easy", 0.0), ("cos", 0.0), ("arc", 2.0)])
def test_margin_ce_and_l2norm(mode, gamma):
    o = ops()
    g = torch.Generator().manual_seed(21)
    B, D, C = 33, 512, 1000
    x = torch.randn(B, D, generator=g, requires_grad=True)
    w = (torch.randn(C, D, generator=g) * 0.05).requires_grad_(True)
    label = torch.randint(0, C, (B,), generator=g)
    s, m = 64.0, (0.5 if mode != "cos" else 0.4)
    logits, loss, cos = _arcface_ref(x, w, label, s, m, mode == "arc_easy", mode == "cos", gamma)
    cos.retain_grad()
    loss.backward()
    cosd = cos.detach().to(DEV)
    lg, lrows, dcos = o.margin_ce(cosd, label.to(DEV), C, mode, s, m, gamma=gamma, grad_scale=1.0 / B, dcos_dtype=torch.float32)
    lossd = o.mean(lrows)
    torch.cuda.synchronize()
    assert torch.allclose(lg.cpu(), logits.detach(), rtol=1e-5, atol=1e-4)
    assert abs(lossd.item() - loss.item()) < 1e-4 * max(1.0, abs(loss.item()))
    assert rel_err(dcos.cpu(), cos.grad) < 1e-4
    # l2 normalisation fwd/bwd
    xd = x.detach().to(DEV)
    xn, xnT, inv = o.l2norm_fwd(xd, torch.float32, want_t=True, ldt=40)
    torch.cuda.synchronize()
    ref = F.normalize(x.detach())
    assert torch.allclose(xn.cpu(), ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(xnT.cpu()[:, :B], ref.t(), rtol=1e-5, atol=1e-6)
    xx = x.detach().clone().requires_grad_(True)
    gg = torch.randn(B, D, generator=g)
    F.normalize(xx).backward(gg)
    dx = o.l2norm_bwd(xd, inv, gg.to(DEV), torch.float32)
    torch.cuda.synchronize()
    assert rel_err(dx.cpu(), xx.grad) < 1e-5


def test_sgd_and_adamw_steps():
    o = ops()
    g = torch.Generator().manual_seed(9)
    n = 10007
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) for _ in range(3)]
    # SGD
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([pr], lr=0.01, momentum=0.9, weight_decay=1e-4)
    pd, md = p0.clone().to(DEV), torch.zeros(n, device=DEV)
    sh = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    for i, gr in enumerate(grads):
        pr.grad = gr.clone()
        opt.step()
        o.sgd_step(pd, gr.to(DEV), md, sh, 0.01, 0.9, 1e-4, first_step=(i == 0))
    torch.cuda.synchronize()
    assert torch.allclose(pd.cpu(), pr.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(sh.float().cpu(), pr.detach().bfloat16().float(), rtol=1e-2, atol=1e-3)
    # AdamW
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3, weight_decay=1e-2)
    pd, m1, v1 = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for i, gr in enumerate(grads):
        pr.grad = gr.clone()
        opt.step()
        o.adamw_step(pd, gr.to(DEV), m1, v1, None, 1e-3, 0.9, 0.999, 1e-8, 1e-2, i + 1)
    torch.cuda.synchronize()
    assert torch.allclose(pd.cpu(), pr.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stem_space_to_depth_exact(dtype):
    """pfr_s2d_input / pfr_s2d_weight / pfr_s2d_wgrad: the 7x7 / stride-2 / pad-3 stem evaluated as a 4x4 stride-1 pad-2
    convolution over the space-to-depth image gives the torch result (forward and weight gradient), fp32 to 1e-6."""
    from pets_face_recognition_amd._hip import lib, dtype_id
    o = ops()
    g = torch.Generator().manual_seed(12)
    N, H, W, C, Co = 4, 64, 48, 3, 64
    x = torch.rand(N, C, H, W, generator=g)
    w = torch.randn(Co, C, 7, 7, generator=g) * 0.1
    xr, wr = x.double(), w.double().requires_grad_(True)
    if dtype == torch.bfloat16:
        xr, wr = x.bfloat16().double(), w.bfloat16().double().requires_grad_(True)
    y = F.conv2d(xr, wr, stride=2, padding=3)
    dy = torch.randn(y.shape, generator=g).double()
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().double()
    y.backward(dy)
    st = torch.cuda.current_stream().cuda_stream
    kp = 8 if dtype == torch.bfloat16 else 4
    Cp = (4 * C + kp - 1) // kp * kp
    xs = torch.empty(N, H // 2, W // 2, Cp, dtype=dtype, device=DEV)
    xd = x.to(DEV)
    lib.pfr_s2d_input(xd.data_ptr(), xs.data_ptr(), dtype_id(dtype), N, C, H, W, Cp, st)
    w_ohwi = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    ws = torch.empty(Co, 4, 4, Cp, dtype=dtype, device=DEV)
    lib.pfr_s2d_weight(w_ohwi.data_ptr(), ws.data_ptr(), dtype_id(dtype), Co, C, Cp, st)
    yh, _ = o.conv2d_fwd(xs, ws, stride=1, pad=2, out_hw=(H // 2, W // 2), out_dtype=torch.float32)
    dyh = dy.float().permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    wsp = torch.empty(lib.pfr_conv2d_wgrad_splits(N * (H // 2) * (W // 2), Co, 16 * Cp) * Co * 16 * Cp, dtype=torch.float32, device=DEV)
    dws = torch.empty(Co, 4, 4, Cp, dtype=torch.float32, device=DEV)
    lib.pfr_conv2d_wgrad(xs.data_ptr(), dyh.data_ptr(), dws.data_ptr(), wsp.data_ptr(), dtype_id(dtype), N, H // 2, W // 2, Cp, Co,
                         4, 4, 1, 2, H // 2, W // 2, Co, 0, 0, 0, 1.0, 0, st)
    dw = torch.full((Co, 7, 7, C), 7.0, dtype=torch.float32, device=DEV)
    lib.pfr_s2d_wgrad(dws.data_ptr(), dw.data_ptr(), Co, C, Cp, 0, st)
    dw2 = dw.clone()
    lib.pfr_s2d_wgrad(dws.data_ptr(), dw2.data_ptr(), Co, C, Cp, 1, st)       # accumulate
    torch.cuda.synchronize()
    t = 1e-6 if dtype == torch.float32 else 1e-5     # bf16 inputs are exactly representable: only the fp32 sum order differs
    assert rel_err(yh.cpu().double().permute(0, 3, 1, 2), y.detach()) < t
    assert rel_err(dw.cpu().double(), wr.grad.permute(0, 2, 3, 1)) < t
    assert torch.allclose(dw2, 2 * dw)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,H,C,Cout", [(1, 16, 64, 128), (1, 12, 96, 64), (3, 16, 64, 64), (3, 10, 32, 96)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_strided_dgrad_parity_classes(dtype, R, H, C, Cout, accumulate):
    """Data gradient of stride-2 convs (3x3/pad 1 and the 1x1 projection shortcut, torchvision resnet downsample) through the
    parity-class path of the persistent kernel: per (oh%2, ow%2) class only the taps that exist; for a 1x1 kernel only class
    (0,0) has any — with accumulate the other classes are not visited, without they store zeros.  Also odd tile counts."""
    o = ops()
    g = torch.Generator().manual_seed(R * 100 + H)
    N, pad = 3, (R - 1) // 2
    x = torch.randn(N, C, H, H, generator=g, requires_grad=True)
    w = torch.randn(Cout, C, R, R, generator=g) / (C * R * R) ** 0.5
    dx0 = torch.randn(N, C, H, H, generator=g)
    OH = (H + 2 * pad - R) // 2 + 1
    dy = torch.randn(N, Cout, OH, OH, generator=g)
    if dtype == torch.bfloat16:
        w, dy, dx0 = w.bfloat16().float(), dy.bfloat16().float(), dx0.bfloat16().float()
    F.conv2d(x, w, stride=2, padding=pad).backward(dy)
    ref = x.grad + (dx0 if accumulate else 0)
    wt = o.weight_dgrad_layout(w.permute(0, 2, 3, 1).contiguous().to(DEV, dtype))
    out = nhwc(dx0).to(DEV, dtype) if accumulate else torch.full((N, H, H, C), float("nan"), device=DEV, dtype=dtype)
    o.conv2d_dgrad(nhwc(dy).to(DEV, dtype), wt, (H, H), 2, pad, R, R, out=out, accumulate=accumulate)
    torch.cuda.synchronize()
    e = rel_err(out.cpu(), nhwc(ref))
    assert e < (1e-2 if dtype == torch.bfloat16 else 1e-5), e


@pytest.mark.parametrize("case", [
    # kind, N, H, C, Cout, R, stride / dilation-log2
    ("fwd", 8, 32, 64, 128, 1, 1), ("fwd", 8, 32, 64, 64, 3, 1), ("fwd", 8, 16, 256, 256, 3, 1), ("fwd", 8, 32, 128, 128, 3, 2),
    ("dgrad", 8, 16, 128, 128, 3, 1), ("dgrad", 8, 32, 256, 64, 1, 0), ("fwd", 20, 16, 96, 192, 1, 1)])
def test_alternative_gemm_kernels_bit_identical(case):
    """The persistent kernel (pfr_igemm_p.hip) must reproduce the one-tile-per-
    workgroup kernel BIT FOR BIT (same MFMA accumulation order) and publish BatchNorm statistics that finalise to the same
    mean / invstd, whatever partial granularity pfr_conv2d_mtile reports for them.  pfr_set_tuning selects the kernel."""
    from pets_face_recognition_amd._hip import lib
    o = ops()
    kind, N, H, C, Co, R, sd = case
    pad = (R - 1) // 2
    g = torch.Generator().manual_seed(H * C + R)
    x = torch.randn(N, H, H, C, generator=g).to(DEV).bfloat16()
    w = (torch.randn(Co, R, R, C, generator=g) / (C * R * R) ** 0.5).to(DEV).bfloat16()
    if kind == "fwd":
        kw = dict(stride=sd, pad=pad, stats=True)
    else:
        kw = dict(stride=1, pad=R - 1 - pad, idil_log2=sd, out_hw=(H << sd, H << sd), stats=False)
    outs = []
    try:
        for knobs in ({"igemm_p": 0}, {"igemm_p": 2, "igemm_ppf": 0}, {"igemm_p": 2, "igemm_ppf": 1}, {"igemm_p": 2, "igemm_ppf": 3}):
            for k, v in knobs.items():
                lib.pfr_set_tuning(k.encode(), v)
            y, part = o.conv2d_fwd(x, w, **kw)
            st = None
            if part is not None:
                M = y.numel() // Co
                OHH = y.shape[1]
                mt = lib.pfr_conv2d_mtile(N, H, H, C, Co, R, R, sd, pad, OHH, OHH, 1, 1, 0)
                st = o.bn_finalize(part, mt, M, None, None, 1e-5, 0.1, None, None)[:2].clone()
            torch.cuda.synchronize()
            outs.append((y.clone(), st))
    finally:
        for k, v in (("igemm_p", 1), ("igemm_ppf", 0)):
            lib.pfr_set_tuning(k.encode(), v)
    for y, st in outs[1:]:
        assert torch.equal(y, outs[0][0])
        if st is not None:
            assert torch.allclose(st, outs[0][1], rtol=2e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_weight_dgrad_layout_batch_ragged_layers(dtype):
    """one launch for many layers (64x64 LDS tiles): wt[ci][R-1-r][S-1-s][o] = w[o][r][s][ci], sizes that are not tile multiples"""
    import struct
    from pets_face_recognition_amd._hip import lib, dtype_id
    g = torch.Generator().manual_seed(3)
    shapes = [(70, 3, 3, 3), (10, 1, 1, 100), (64, 1, 1, 64), (130, 3, 3, 65), (1, 1, 1, 1), (256, 4, 4, 16)]
    ws = [torch.randn(s, generator=g).to(DEV, dtype) for s in shapes]
    wts = [torch.full((s[3], s[1], s[2], s[0]), float("nan"), device=DEV, dtype=dtype) for s in shapes]
    raw = b"".join(struct.pack("<QQiiii", w.data_ptr(), t.data_ptr(), s[0], s[1], s[2], s[3]) for w, t, s in zip(ws, wts, shapes))
    tab = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(DEV)
    lib.pfr_weight_dgrad_layout_batch(tab.data_ptr(), len(shapes), dtype_id(dtype), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for w, t in zip(ws, wts):
        assert torch.equal(t, w.flip(1, 2).permute(3, 1, 2, 0).contiguous())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(5, 3, 7, 9, 8), (4, 3, 6, 6, 4), (96, 3, 4, 4, 8), (7, 20, 2, 2, 24), (3, 5, 3, 3, 7), (2, 16, 5, 5, 16)])
def test_nchw_to_nhwc_channel_padding(dtype, shape):
    """fp32 NCHW → compute-dtype NHWC with the channel dimension zero-padded to Cp (16-byte-store and scalar paths)"""
    from pets_face_recognition_amd._hip import lib, dtype_id
    N, C, H, W, Cp = shape
    if dtype == torch.float32 and Cp == 8 and C == 3:
        Cp = 4
    x = torch.randn(N, C, H, W, generator=torch.Generator().manual_seed(5)).to(DEV)
    y = torch.full((N, H, W, Cp), float("nan"), device=DEV, dtype=dtype)
    lib.pfr_nchw_to_nhwc(x.data_ptr(), y.data_ptr(), dtype_id(dtype), N, C, H, W, Cp, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = torch.zeros(N, H, W, Cp, device=DEV, dtype=dtype)
    ref[..., :C] = x.permute(0, 2, 3, 1).to(dtype)
    assert torch.equal(y, ref)


@pytest.mark.parametrize("case", [
    # (N, H, C, Cout, stride): weight-stationary streaming kernel geometries (pfr_sconv.hip), incl. ragged row counts
    (2, 56, 64, 64, 1), (2, 56, 64, 256, 1), (2, 56, 256, 64, 1), (3, 28, 128, 512, 1), (2, 28, 512, 128, 1),
    (5, 14, 256, 1024, 1), (2, 56, 256, 128, 1), (2, 56, 256, 512, 2), (1, 7, 128, 256, 1), (3, 5, 64, 64, 1), (7, 9, 256, 64, 1),
    (1, 3, 64, 256, 1),
])
def test_streaming_1x1_kernel_bit_identical(case):
    """pfr_sconv.hip (persistent, weight panel resident in LDS, per-wave DMA rings) must reproduce the tile kernel BIT FOR BIT and
    publish BatchNorm statistics that finalise to the same mean / invstd at the granularity pfr_conv2d_mtile reports."""
    from pets_face_recognition_amd._hip import lib
    o = ops()
    N, H, C, Co, sd = case
    g = torch.Generator().manual_seed(H * C + Co)
    x = torch.randn(N, H, H, C, generator=g).to(DEV).bfloat16()
    w = (torch.randn(Co, 1, 1, C, generator=g) / C ** 0.5).to(DEV).bfloat16()
    outs = []
    try:
        for mode in (0, 2):
            lib.pfr_set_tuning(b"sconv", mode)
            for stats in (True, False):
                y, part = o.conv2d_fwd(x, w, stride=sd, pad=0, stats=stats)
                st = None
                if part is not None:
                    M = y.numel() // Co
                    mt = lib.pfr_conv2d_mtile(N, H, H, C, Co, 1, 1, sd, 0, y.shape[1], y.shape[2], 1, 1, 0)
                    st = o.bn_finalize(part, mt, M, None, None, 1e-5, 0.1, None, None)[:2].clone()
                torch.cuda.synchronize()
                outs.append((y.clone(), st))
    finally:
        lib.pfr_set_tuning(b"sconv", 1)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=sd).permute(0, 2, 3, 1)
    assert (outs[0][0].float() - ref).abs().max() <= 2e-2 * ref.abs().max()
    for y, st in outs[1:]:
        assert torch.equal(y, outs[0][0])
    assert torch.allclose(outs[2][1], outs[0][1], rtol=2e-4, atol=1e-5)


@pytest.mark.parametrize("case", [(2, 56, 64, 256), (3, 28, 128, 512), (5, 14, 256, 1024), (2, 56, 256, 64), (1, 9, 64, 64), (4, 28, 512, 128)])
def test_streaming_1x1_residual_join_bit_identical(case):
    """pfr_conv2d_dgrad_join (dx = dgrad(dy) + res where the block's ReLU bit is set) through the streaming kernel's join variant
    (residual / mask rows prefetched by hand-counted loads) vs the tile kernel, bit for bit, and vs torch."""
    from pets_face_recognition_amd._hip import lib
    N, H, C, Co = case
    g = torch.Generator().manual_seed(H * C + Co + 1)
    dy = torch.randn(N, H, H, C, generator=g).to(DEV).bfloat16()
    wt = (torch.randn(Co, 1, 1, C, generator=g) / C ** 0.5).to(DEV).bfloat16()
    res = torch.randn(N, H, H, Co, generator=g).to(DEV).bfloat16()
    mask = torch.randint(0, 256, (N * H * H, Co // 8), generator=g, dtype=torch.uint8).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    try:
        for mode in (0, 2):
            lib.pfr_set_tuning(b"sconv", mode)
            dx = torch.full((N, H, H, Co), float("nan"), device=DEV, dtype=torch.bfloat16)
            lib.pfr_conv2d_dgrad_join(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), 1, N, H, H, C, Co, 1, 1, 0, 0, H, H, res.data_ptr(),
                                      mask.data_ptr(), st)
            torch.cuda.synchronize()
            outs.append(dx)
    finally:
        lib.pfr_set_tuning(b"sconv", 1)
    bits = ((mask.unsqueeze(-1) >> torch.arange(8, device=DEV, dtype=torch.uint8)) & 1).reshape(N, H, H, Co).float()
    core = (dy.float().reshape(-1, C) @ wt.float().reshape(Co, C).t()).bfloat16().float().reshape(N, H, H, Co)
    ref = (core + res.float() * bits).bfloat16()
    assert torch.equal(outs[0], outs[1])
    assert (outs[1].float() - ref.float()).abs().max() <= 2e-2 * ref.float().abs().max()


@pytest.mark.parametrize("case", [(2, 56, 56), (3, 16, 16), (1, 8, 8), (2, 12, 16), (5, 24, 40), (1, 4, 8), (9, 28, 32)])
def test_halo_staged_3x3_kernel_bit_identical(case):
    """pfr_sconv3.hip (64 -> 64 channels, 3x3 / stride 1 / pad 1: weights resident in LDS, one 6x10-pixel halo tile per 4x8 patch,
    nine taps read from it) must reproduce the implicit-GEMM tile kernel BIT FOR BIT — forward with BatchNorm statistics, forward
    without, and as data gradient (tap-flipped weights) — and match torch."""
    from pets_face_recognition_amd._hip import lib
    o = ops()
    N, H, W = case
    g = torch.Generator().manual_seed(H * W + N)
    x = torch.randn(N, H, W, 64, generator=g).to(DEV).bfloat16()
    w = (torch.randn(64, 3, 3, 64, generator=g) / 24.0).to(DEV).bfloat16()
    wt = o.weight_dgrad_layout(w)
    outs = []
    try:
        for mode in (0, 2):
            lib.pfr_set_tuning(b"sconv", mode)
            y, part = o.conv2d_fwd(x, w, stride=1, pad=1, stats=True)
            mt = lib.pfr_conv2d_mtile(N, H, W, 64, 64, 3, 3, 1, 1, H, W, 1, 1, 0)
            st = o.bn_finalize(part, mt, N * H * W, None, None, 1e-5, 0.1, None, None)[:2].clone()
            y2, _ = o.conv2d_fwd(x, w, stride=1, pad=1, stats=False)
            dx = o.conv2d_dgrad(x, wt, (H, W), 1, 1, 3, 3)
            torch.cuda.synchronize()
            outs.append((y.clone(), st, y2.clone(), dx.clone()))
    finally:
        lib.pfr_set_tuning(b"sconv", 1)
    xf, wf = x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xf, wf, padding=1).permute(0, 2, 3, 1)
    assert (outs[0][0].float() - ref).abs().max() <= 2e-2 * ref.abs().max()
    refd = torch.nn.functional.conv_transpose2d(xf, wf, padding=1).permute(0, 2, 3, 1)
    assert (outs[0][3].float() - refd).abs().max() <= 2e-2 * refd.abs().max()
    assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][2], outs[0][2]) and torch.equal(outs[1][3], outs[0][3])
    assert torch.equal(outs[1][0], outs[1][2])
    assert torch.allclose(outs[1][1], outs[0][1], rtol=2e-4, atol=1e-5)


@pytest.mark.parametrize("case", [(16, 112, 112), (64, 32, 32), (176, 16, 24), (5, 112, 112), (70, 28, 40), (2100, 4, 8), (300, 8, 8), (7, 60, 88)])
def test_halo_staged_stem_kernel_bit_identical(case):
    """pfr_sstem.hip (the space-to-depth stem: 4x4 / stride 1 / pad 2 over 16 channels -> 64; weights resident in LDS, one 7x11-pixel halo tile
    per 4x8 patch, a tap = one k-group, permuted pixel <-> MFMA-column map) must reproduce the implicit-GEMM tile kernel BIT FOR BIT — with
    BatchNorm statistics (finalised mean / invstd equal to 2e-4) and without — and match torch; the inference epilogue (bias + ReLU) within
    the one-extra-bf16-rounding bound of test_halo_staged_3x3_inference_epilogue.  Small cases fall back to the tile kernel (too few patches):
    the knob must then change nothing."""
    from pets_face_recognition_amd._hip import lib
    o = ops()
    N, H, W = case
    g = torch.Generator().manual_seed(H * W + N)
    x = torch.randn(N, H, W, 16, generator=g).to(DEV).bfloat16()
    x[..., 12:] = 0                                          # the space-to-depth image's padding channels
    w = (torch.randn(64, 4, 4, 16, generator=g) / 14.0).to(DEV).bfloat16()
    bias = torch.randn(64, generator=g).to(DEV)
    outs = []
    try:
        for mode in (0, 1):
            lib.pfr_set_tuning(b"sstem", mode)
            y, part = o.conv2d_fwd(x, w, stride=1, pad=2, out_hw=(H, W), stats=True)
            mt = lib.pfr_conv2d_mtile(N, H, W, 16, 64, 4, 4, 1, 2, H, W, 1, 1, 0)
            st = o.bn_finalize(part, mt, N * H * W, None, None, 1e-5, 0.1, None, None)[:2].clone()
            y2, _ = o.conv2d_fwd(x, w, stride=1, pad=2, out_hw=(H, W), stats=False)
            yi, _ = o.conv2d_fwd(x, w, stride=1, pad=2, out_hw=(H, W), bias=bias, out_relu=True)
            torch.cuda.synchronize()
            outs.append((y.clone(), st, y2.clone(), yi.clone()))
    finally:
        lib.pfr_set_tuning(b"sstem", 1)
    xf, wf = x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2)
    conv = torch.nn.functional.conv2d(torch.nn.functional.pad(xf, (2, 1, 2, 1)), wf).permute(0, 2, 3, 1)
    assert conv.shape == outs[0][0].shape
    assert (outs[0][0].float() - conv).abs().max() <= 2e-2 * conv.abs().max()
    assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][2], outs[0][2])
    assert torch.equal(outs[1][0], outs[1][2])
    assert torch.allclose(outs[1][1], outs[0][1], rtol=2e-4, atol=1e-5)
    ref = torch.relu(conv + bias)
    bound = 2.0 ** -8 * ref.abs() + 2.0 ** -8 * conv.abs() + 1e-6
    assert bool(((outs[1][3].float() - ref).abs() <= bound).all())
    assert bool(((outs[0][3].float() - ref).abs() <= bound).all())
    assert (outs[1][3] != outs[0][3]).float().mean().item() < 0.35


@pytest.mark.parametrize("case", [
    # (N, H, C, Cout, stride, residual, relu): the BN-folded inference convolutions the streaming kernel takes
    (8, 56, 64, 256, 1, True, True), (8, 56, 256, 64, 1, False, True), (8, 56, 64, 64, 1, False, True), (16, 28, 512, 128, 1, False, True),
    (16, 28, 128, 512, 1, True, True), (32, 14, 256, 1024, 1, True, True), (8, 56, 256, 512, 2, False, False), (3, 9, 64, 256, 1, True, False),
])
def test_streaming_1x1_inference_epilogue(case):
    """pfr_sconv.hip, inference epilogue (y = relu?(conv + bias [+ residual]), the BN-folded eval plan): adds to the bf16-rounded
    convolution result, so it may differ from the tile kernel's fp32 epilogue by ONE bf16 rounding of the pre-activation — checked
    against both the tile kernel and an fp32 torch reference with exactly that bound."""
    from pets_face_recognition_amd._hip import lib
    o = ops()
    N, H, C, Co, sd, has_res, relu = case
    g = torch.Generator().manual_seed(H * C + Co + 7)
    x = torch.randn(N, H, H, C, generator=g).to(DEV).bfloat16()
    w = (torch.randn(Co, 1, 1, C, generator=g) / C ** 0.5).to(DEV).bfloat16()
    bias = torch.randn(Co, generator=g).to(DEV)
    OH = (H - 1) // sd + 1
    res = torch.randn(N, OH, OH, Co, generator=g).to(DEV).bfloat16() if has_res else None
    outs = []
    try:
        for mode in (0, 2):
            lib.pfr_set_tuning(b"sconv", mode)
            y, _ = o.conv2d_fwd(x, w, stride=sd, pad=0, bias=bias, residual=res, out_relu=relu)
            torch.cuda.synchronize()
            outs.append(y.clone())
    finally:
        lib.pfr_set_tuning(b"sconv", 1)
    conv = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=sd).permute(0, 2, 3, 1)
    pre = conv + bias + (res.float() if has_res else 0)
    ref = torch.relu(pre) if relu else pre
    # bounds: output rounding (2^-9 relative of the result) + one rounding of the convolution value (2^-9 of |conv|) + fp32 noise
    bound = 2.0 ** -8 * ref.abs() + 2.0 ** -8 * conv.abs() + 1e-6
    assert bool(((outs[1].float() - ref).abs() <= bound).all())
    assert bool(((outs[0].float() - ref).abs() <= bound).all())
    assert bool(((outs[1].float() - outs[0].float()).abs() <= 2 * bound).all())
    assert (outs[1] != outs[0]).float().mean().item() < 0.35     # most elements agree exactly; the rest by one rounding step


@pytest.mark.parametrize("case", [(2, 56, 56, True), (3, 16, 16, True), (1, 4, 8, False), (5, 24, 40, True), (16, 56, 56, True)])
def test_halo_staged_3x3_inference_epilogue(case):
    """pfr_sconv3.hip, inference epilogue (y = relu?(conv + bias), BN-folded eval plan) vs the tile kernel and an fp32 torch
    reference, with the one-extra-bf16-rounding bound of test_streaming_1x1_inference_epilogue."""
    from pets_face_recognition_amd._hip import lib
    o = ops()
    N, H, W, relu = case
    g = torch.Generator().manual_seed(H * W + N + 3)
    x = torch.randn(N, H, W, 64, generator=g).to(DEV).bfloat16()
    w = (torch.randn(64, 3, 3, 64, generator=g) / 24.0).to(DEV).bfloat16()
    bias = torch.randn(64, generator=g).to(DEV)
    outs = []
    try:
        for mode in (0, 2):
            lib.pfr_set_tuning(b"sconv", mode)
            y, _ = o.conv2d_fwd(x, w, stride=1, pad=1, bias=bias, out_relu=relu)
            torch.cuda.synchronize()
            outs.append(y.clone())
    finally:
        lib.pfr_set_tuning(b"sconv", 1)
    conv = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    pre = conv + bias
    ref = torch.relu(pre) if relu else pre
    bound = 2.0 ** -8 * ref.abs() + 2.0 ** -8 * conv.abs() + 1e-6
    assert bool(((outs[1].float() - ref).abs() <= bound).all())
    assert bool(((outs[0].float() - ref).abs() <= bound).all())
    assert (outs[1] != outs[0]).float().mean().item() < 0.35


@pytest.mark.parametrize("case", [
    # (N, H, C = channels of dy, Co = channels of dx, join): streaming data gradient + BatchNorm-backward sums (pfr_sconv.hip EP 4 / 5)
    (2, 56, 256, 64, False), (3, 28, 512, 128, False), (2, 56, 64, 256, True), (3, 28, 128, 512, True), (5, 14, 256, 1024, True),
    (1, 9, 64, 64, False), (4, 23, 128, 256, True), (2, 56, 64, 256, "two"), (5, 14, 256, 1024, "two"), (3, 11, 128, 512, "two"),
    (2, 56, 128, 256, "inplace"), (3, 28, 256, 512, "inplace"), (5, 14, 512, 1024, "inplace"),
    (2, 56, 128, 256, "sub"), (3, 28, 256, 512, "sub"), (5, 14, 512, 1024, "sub"), (3, 10, 64, 64, "sub"),
])
def test_streaming_1x1_dgrad_with_bn_backward_sums(case):
    """pfr_conv2d_dgrad_bn in streaming mode (pfr_set_tuning("bnb", 2)): dx must be BIT-identical to the plain data gradient / join,
    and the partial sums must finalise (pfr_bn_bwd_finalize) to the dgamma / dbeta / coefficients of the separate
    pfr_bn_bwd_reduce pass over the finished gradient — recomputed mask (inner BNs) and bit mask (block-output BN with the join)."""
    from pets_face_recognition_amd._hip import lib
    N, H, C, Co, join = case
    two = join == "two"      # + the projection-shortcut BN of the previous block: same gradient, same bit mask, its own input
    inplace = join == "inplace"   # res = dx itself, no residual mask: the main branch adds to what the projection shortcut left there
    sub = join == "sub"           # res = the COMPACT gradient of a stride-2 projection shortcut, added at even (oh, ow) (pfr_conv2d_dgrad_bn_sub)
    join = bool(join)
    g = torch.Generator().manual_seed(H * C + Co + 11)
    M = N * H * H
    dy = torch.randn(N, H, H, C, generator=g).to(DEV).bfloat16()
    wt = (torch.randn(Co, 1, 1, C, generator=g) / C ** 0.5).to(DEV).bfloat16()
    bnx = torch.randn(N, H, H, Co, generator=g).to(DEV).bfloat16()
    coef = torch.stack([torch.randn(Co, generator=g) * 0.1, torch.rand(Co, generator=g) + 0.5, torch.rand(Co, generator=g) + 0.5,
                        torch.randn(Co, generator=g) * 0.3]).to(DEV).contiguous()     # mean, invstd, scale, shift
    gamma = (torch.rand(Co, generator=g) + 0.5).to(DEV)
    res = torch.randn(N, H, H, Co, generator=g).to(DEV).bfloat16() if join else None
    rmask = torch.randint(0, 256, (M, Co // 8), generator=g, dtype=torch.uint8).to(DEV) if (join and not inplace and not sub) else None
    comp = torch.randn(N, H // 2, H // 2, Co, generator=g).to(DEV).bfloat16() if sub else None
    bmask = torch.randint(0, 256, (M, Co // 8), generator=g, dtype=torch.uint8).to(DEV) if join else None
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: 0 if t is None else t.data_ptr()
    try:
        lib.pfr_set_tuning(b"sconv", 2)
        # reference: plain gradient (streaming kernel, already proven bit-identical to the tile kernel) + separate reduce
        dx0 = torch.full((N, H, H, Co), float("nan"), device=DEV, dtype=torch.bfloat16)
        if sub:
            lib.pfr_conv2d_fwd(dy.data_ptr(), wt.data_ptr(), dx0.data_ptr(), 1, 1, N, H, H, C, Co, 1, 1, 1, 0, 0, H, H, Co, 0, 0, 0, 0, 0, 0, 0, 0, st)
            up = torch.zeros(N, H, H, Co, device=DEV)
            up[:, 0::2, 0::2] = comp.float()
            dx0 = (dx0.float() + up).bfloat16()
        elif inplace:
            dx0.copy_(res)
            lib.pfr_conv2d_fwd(dy.data_ptr(), wt.data_ptr(), dx0.data_ptr(), 1, 1, N, H, H, C, Co, 1, 1, 1, 0, 0, H, H, Co, 0, 0, 1, 0, 0, 0, 0, 0, st)
        elif join:
            lib.pfr_conv2d_dgrad_join(dy.data_ptr(), wt.data_ptr(), dx0.data_ptr(), 1, N, H, H, C, Co, 1, 1, 0, 0, H, H, res.data_ptr(), rmask.data_ptr(), st)
        else:
            lib.pfr_conv2d_fwd(dy.data_ptr(), wt.data_ptr(), dx0.data_ptr(), 1, 1, N, H, H, C, Co, 1, 1, 1, 0, 0, H, H, Co, 0, 0, 0, 0, 0, 0, 0, 0, st)
        nb = lib.pfr_colreduce_blocks(Co, 1, M)
        part0 = torch.zeros(nb, 2, Co, device=DEV)
        lib.pfr_bn_bwd_reduce(dx0.data_ptr(), P(bmask), bnx.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(),
                              coef[3].data_ptr(), 3 if join else 2, 1, M, Co, part0.data_ptr(), st)
        fin0 = [torch.empty(Co, device=DEV), torch.empty(Co, device=DEV), torch.empty(3, Co, device=DEV)]
        lib.pfr_bn_bwd_finalize(part0.data_ptr(), nb, Co, float(M), gamma.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(),
                                fin0[0].data_ptr(), fin0[1].data_ptr(), fin0[2].data_ptr(), 0, st)
        bnx2 = torch.randn(N, H, H, Co, generator=g).to(DEV).bfloat16() if two else None
        coef2 = torch.stack([torch.randn(Co, generator=g) * 0.1, torch.rand(Co, generator=g) + 0.5, torch.ones(Co), torch.zeros(Co)]).to(DEV).contiguous() if two else None
        if two:
            part02 = torch.zeros(nb, 2, Co, device=DEV)
            lib.pfr_bn_bwd_reduce(dx0.data_ptr(), P(bmask), bnx2.data_ptr(), coef2[0].data_ptr(), coef2[1].data_ptr(), coef2[2].data_ptr(),
                                  coef2[3].data_ptr(), 3, 1, M, Co, part02.data_ptr(), st)
            fin02 = [torch.empty(Co, device=DEV), torch.empty(Co, device=DEV), torch.empty(3, Co, device=DEV)]
            lib.pfr_bn_bwd_finalize(part02.data_ptr(), nb, Co, float(M), gamma.data_ptr(), coef2[0].data_ptr(), coef2[1].data_ptr(),
                                    fin02[0].data_ptr(), fin02[1].data_ptr(), fin02[2].data_ptr(), 0, st)
        # fused
        lib.pfr_set_tuning(b"bnb", 2)
        np_ = lib.pfr_conv2d_dgrad_bn_parts(1, N, H, H, C, Co, 1, 1, 0, H, H)
        assert 0 < np_ <= 256
        part1 = torch.full((np_, 2, Co), float("nan"), device=DEV)
        part12 = torch.full((np_, 2, Co), float("nan"), device=DEV) if two else None
        dx1 = torch.full((N, H, H, Co), float("nan"), device=DEV, dtype=torch.bfloat16)
        for _ in range(2):
            if sub:
                lib.pfr_conv2d_dgrad_bn_sub(dy.data_ptr(), wt.data_ptr(), dx1.data_ptr(), 1, N, H, H, C, Co, H, H, comp.data_ptr(),
                                            bnx.data_ptr(), coef.data_ptr(), bmask.data_ptr(), part1.data_ptr(), st)
                continue
            if inplace:
                dx1.copy_(res)
            lib.pfr_conv2d_dgrad_bn(dy.data_ptr(), wt.data_ptr(), dx1.data_ptr(), 1, N, H, H, C, Co, 1, 1, 0, 0, H, H,
                                    dx1.data_ptr() if inplace else P(res), P(rmask), 0,
                                    bnx.data_ptr(), coef.data_ptr(), P(bmask), part1.data_ptr(), P(bnx2), P(coef2), P(part12), st)
        if two:
            fin12 = [torch.empty(Co, device=DEV), torch.empty(Co, device=DEV), torch.empty(3, Co, device=DEV)]
            lib.pfr_bn_bwd_finalize(part12.data_ptr(), np_, Co, float(M), gamma.data_ptr(), coef2[0].data_ptr(), coef2[1].data_ptr(),
                                    fin12[0].data_ptr(), fin12[1].data_ptr(), fin12[2].data_ptr(), 0, st)
        fin1 = [torch.empty(Co, device=DEV), torch.empty(Co, device=DEV), torch.empty(3, Co, device=DEV)]
        lib.pfr_bn_bwd_finalize(part1.data_ptr(), np_, Co, float(M), gamma.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(),
                                fin1[0].data_ptr(), fin1[1].data_ptr(), fin1[2].data_ptr(), 0, st)
        torch.cuda.synchronize()
    finally:
        lib.pfr_set_tuning(b"sconv", 1)
        lib.pfr_set_tuning(b"bnb", 0)
    assert torch.equal(dx0, dx1)
    sg, sgx = part0[:, 0].abs().sum(0) + 1e-3, part0[:, 1].abs().sum(0) + 1e-3
    assert ((fin1[1] - fin0[1]).abs() / sg).max().item() < 5e-5          # dbeta = sum g
    assert ((fin1[0] - fin0[0]).abs() / sgx).max().item() < 5e-5         # dgamma = sum g xhat
    assert torch.allclose(fin1[2], fin0[2], rtol=3e-4, atol=1e-5 * float(fin0[2].abs().max()))
    if two:
        sgx2 = part02[:, 1].abs().sum(0) + 1e-3
        assert ((fin12[1] - fin02[1]).abs() / sg).max().item() < 5e-5
        assert ((fin12[0] - fin02[0]).abs() / sgx2).max().item() < 5e-5
        assert torch.allclose(fin12[2], fin02[2], rtol=3e-4, atol=1e-5 * float(fin02[2].abs().max()))


@pytest.mark.parametrize("case", [
    # (N, H, C, Cout): 1x1 / stride-1 weight gradients through the streaming kernel (pfr_wgrad.hip swgrad_kernel), incl. ragged M
    (16, 56, 64, 256), (16, 56, 256, 64), (8, 56, 64, 64), (9, 28, 128, 512), (9, 28, 512, 128), (33, 14, 256, 1024), (16, 28, 256, 256),
    (3, 55, 64, 256), (5, 37, 128, 128),
])
def test_streaming_1x1_weight_gradient(case):
    """dw = dy^T x of a 1x1 convolution through the barrier-free streaming kernel (per-wave LDS-DMA rings, transposing LDS reads,
    wave groups summed through LDS) vs the tile kernel and an fp32 torch reference; deterministic from launch to launch."""
    from pets_face_recognition_amd._hip import lib
    o = ops()
    N, H, C, Co = case
    g = torch.Generator().manual_seed(H * C + Co + 13)
    x = torch.randn(N, H, H, C, generator=g).to(DEV).bfloat16()
    dy = torch.randn(N, H, H, Co, generator=g).to(DEV).bfloat16()
    outs = []
    try:
        for mode in (0, 2, 2):
            lib.pfr_set_tuning(b"swgrad", mode)
            dw = o.conv2d_wgrad(x, dy, 1, 1, 1, 0)
            torch.cuda.synchronize()
            outs.append(dw.clone())
    finally:
        lib.pfr_set_tuning(b"swgrad", 1)
    ref = (dy.float().reshape(-1, Co).t().double() @ x.float().reshape(-1, C).double()).float().reshape(Co, 1, 1, C)
    scale = ref.abs().max()
    assert (outs[0] - ref).abs().max() <= 2e-5 * scale + 1e-3
    assert (outs[1] - ref).abs().max() <= 2e-5 * scale + 1e-3
    assert torch.equal(outs[1], outs[2])


@pytest.mark.parametrize("case", [(2, 56, 256, 64), (3, 28, 128, 512), (1, 9, 64, 64), (4, 14, 256, 1024)])
def test_streaming_1x1_accumulate_bit_identical(case):
    """y += conv through the streaming kernel (the join with res = y itself, no mask) vs the tile kernel's accumulate epilogue"""
    from pets_face_recognition_amd._hip import lib
    o = ops()
    N, H, C, Co = case
    g = torch.Generator().manual_seed(H * C + Co + 17)
    x = torch.randn(N, H, H, C, generator=g).to(DEV).bfloat16()
    w = (torch.randn(Co, 1, 1, C, generator=g) / C ** 0.5).to(DEV).bfloat16()
    y0 = torch.randn(N, H, H, Co, generator=g).to(DEV).bfloat16()
    outs = []
    try:
        for mode in (0, 2):
            lib.pfr_set_tuning(b"sconv", mode)
            y = y0.clone()
            o.conv2d_fwd(x, w, stride=1, pad=0, out=y, accumulate=True)
            torch.cuda.synchronize()
            outs.append(y)
    finally:
        lib.pfr_set_tuning(b"sconv", 1)
    assert torch.equal(outs[0], outs[1])
    ref = (x.float().reshape(-1, C) @ w.float().reshape(Co, C).t()).bfloat16().float().reshape(N, H, H, Co) + y0.float()
    assert (outs[1].float() - ref).abs().max() <= 2e-2 * ref.abs().max()


def test_statistics_partials_agree_with_the_query_at_the_2gib_input_boundary():
    """ADVICE r3 (pfr_igemm.hip:804): a stride-2 1x1 conv whose OUTPUT-side extents pass the streaming kernel's 2 GiB test while its
    INPUT extent (4x the rows) does not — layer2.0.downsample at a per-GPU batch of ≈ 1.4 k.  pfr_conv2d_mtile and the launch must
    agree on the kernel (one shared geometry test): the partial rows the query promises are the rows the launch writes (canary row
    behind them untouched) and they finalise to the statistics of the output.  And a statistics launch with post-ops that only the
    tile kernel can take fails with PFR_ERR_ARG instead of writing a different number of partial rows."""
    from pets_face_recognition_amd._hip import lib, PfrError
    o = ops()
    N, H, C, Co = 1340, 56, 256, 512                      # N*H*H*C*2 = 2.15e9 >= 2^31;  M*C*2 = 5.4e8
    assert N * H * H * C * 2 >= 2 ** 31 and N * (H // 2) ** 2 * max(C, Co) * 2 < 2 ** 31
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(N, H, H, C, generator=g, device=DEV, dtype=torch.bfloat16)
    w = (torch.randn(Co, 1, 1, C, generator=g, device=DEV) / C ** 0.5).bfloat16()
    M = N * (H // 2) ** 2
    mt = lib.pfr_conv2d_mtile(N, H, H, C, Co, 1, 1, 2, 0, H // 2, H // 2, 1, 1, 0)
    nt = (M + mt - 1) // mt
    buf = torch.full((nt + 1, 2, Co), 12345.0, dtype=torch.float32, device=DEV)
    y, part = o.conv2d_fwd(x, w, stride=2, pad=0, stats=True, stats_buf=buf)
    torch.cuda.synchronize()
    assert torch.all(buf[nt] == 12345.0), "the launch wrote more partial rows than pfr_conv2d_mtile promised"
    assert not torch.any(buf[:nt] == 12345.0), "the launch wrote fewer partial rows than pfr_conv2d_mtile promised"
    st = o.bn_finalize(buf[:nt], mt, M, None, None, 1e-5, 0.1, None, None)
    yf = y.float().reshape(M, Co)
    mean, var = yf.mean(0), yf.var(0, unbiased=False)
    assert torch.allclose(st[0], mean, rtol=1e-3, atol=2e-3)
    assert torch.allclose(st[1], (var + 1e-5).rsqrt(), rtol=2e-3)
    # just below the boundary the streaming kernel takes the same layer (its row range is the granularity)
    mt_small = lib.pfr_conv2d_mtile(256, H, H, C, Co, 1, 1, 2, 0, H // 2, H // 2, 1, 1, 0)
    assert mt_small != mt or mt in (64, 128, 256)
    # statistics + ReLU post-op on a geometry whose query answers the streaming granularity: loud error, nothing written
    xs = x[:8]
    mts = lib.pfr_conv2d_mtile(8, H, H, C, Co, 1, 1, 1, 0, H, H, 1, 1, 0)
    if mts not in (64, 128, 256):
        with pytest.raises(PfrError):
            o.conv2d_fwd(xs, w, stride=1, pad=0, stats=True, out_relu=True)


@pytest.mark.parametrize("case", [(2, 56, 64, 256, False), (3, 28, 128, 512, True), (2, 56, 64, 256, True), (1, 9, 64, 256, False),
                                  (5, 14, 256, 1024, False), (7, 9, 128, 512, True)])
def test_recompute_form_of_the_last_conv_is_bit_identical(case):
    """pfr_conv1x1_stats + pfr_conv1x1_bn_tail (csrc/pfr_sconv.hip EP 9 / 10 / 11): the block tail relu(bn3(conv3(z)) + shortcut) with
    conv3's output never stored must give the BITS of pfr_conv2d_fwd (statistics) + pfr_bn_act_mask: same output, same ReLU bit mask,
    statistics partials that finalise to the same coefficients — identity and projection (second BN on the shortcut) forms, ragged
    row counts."""
    from pets_face_recognition_amd._hip import lib
    o = ops()
    N, H, C, Co, proj = case
    g = torch.Generator().manual_seed(H * C + Co + proj)
    z = torch.relu(torch.randn(N, H, H, C, generator=g)).to(DEV).bfloat16()
    w = (torch.randn(Co, 1, 1, C, generator=g) / C ** 0.5).to(DEV).bfloat16()
    res = torch.randn(N, H, H, Co, generator=g).to(DEV).bfloat16()
    a2 = (1 + 0.1 * torch.randn(Co, generator=g)).to(DEV) if proj else None
    b2 = (0.1 * torch.randn(Co, generator=g)).to(DEV) if proj else None
    gamma = (1 + 0.1 * torch.randn(Co, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(Co, generator=g)).to(DEV)
    M = N * H * H
    try:
        lib.pfr_set_tuning(b"sconv", 2)
        mt = lib.pfr_conv1x1_tail_mtile(1, N, H, H, C, Co)
        assert mt > 0 and mt == lib.pfr_conv2d_mtile(N, H, H, C, Co, 1, 1, 1, 0, H, H, 1, 1, 0)
        st = torch.cuda.current_stream().cuda_stream
        # stored form
        y, part = o.conv2d_fwd(z, w, stats=True)
        coef = o.bn_finalize(part, mt, M, gamma, beta, 1e-5, 0.1, None, None).clone()
        out_ref = torch.empty_like(y)
        mask_ref = torch.zeros(M, Co // 8, dtype=torch.uint8, device=DEV)
        lib.pfr_bn_act_mask(y.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), res.data_ptr(), a2.data_ptr() if proj else 0,
                            b2.data_ptr() if proj else 0, out_ref.data_ptr(), mask_ref.data_ptr(), 1, M, Co, 1, st)
        # recompute form
        nt = (M + mt - 1) // mt
        part2 = torch.full((nt + 1, 2, Co), 777.0, dtype=torch.float32, device=DEV)
        lib.pfr_conv1x1_stats(z.data_ptr(), w.data_ptr(), 1, N, H, H, C, Co, part2.data_ptr(), st)
        coef2 = o.bn_finalize(part2[:nt], mt, M, gamma, beta, 1e-5, 0.1, None, None).clone()
        out = torch.empty_like(y)
        mask = torch.zeros(M, Co // 8, dtype=torch.uint8, device=DEV)
        lib.pfr_conv1x1_bn_tail(z.data_ptr(), w.data_ptr(), out.data_ptr(), mask.data_ptr(), 1, N, H, H, C, Co, coef2[2].data_ptr(),
                                coef2[3].data_ptr(), res.data_ptr(), a2.data_ptr() if proj else 0, b2.data_ptr() if proj else 0, st)
        torch.cuda.synchronize()
    finally:
        lib.pfr_set_tuning(b"sconv", 1)
    assert torch.all(part2[nt] == 777.0)
    assert torch.equal(part2[:nt], part) and torch.equal(coef2, coef)
    assert torch.equal(out, out_ref) and torch.equal(mask, mask_ref)
    assert out.float().abs().sum() > 0 and mask.sum() > 0


@pytest.mark.parametrize("M,Q", [(8192, 64), (100357, 64), (4099, 128), (25088, 128), (200704, 128)])
def test_gram_matrix_and_column_sums_in_one_pass(M, Q):
    """pfr_gram_colsum (csrc/pfr_wgrad.hip gram_kernel): G2 = XᵀX and the column sums of X [M][Q] (bf16) from ONE streaming pass,
    against fp64 torch on the same bf16 values; deterministic (two launches give the same bits); ragged row counts."""
    from pets_face_recognition_amd._hip import lib
    g = torch.Generator().manual_seed(M + Q)
    x = torch.relu(torch.randn(M, Q, generator=g)).bfloat16().to(DEV)
    nws = lib.pfr_gram_ws_floats(M, Q)
    assert nws > 0 and lib.pfr_gram_ws_floats(M, 96) == 0
    ws = torch.empty(nws, dtype=torch.float32, device=DEV)
    outs = []
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        out = torch.full((Q * Q + Q + 8,), -3.0, dtype=torch.float32, device=DEV)
        lib.pfr_gram_colsum(x.data_ptr(), 1, M, Q, out.data_ptr(), ws.data_ptr(), st)
        torch.cuda.synchronize()
        outs.append(out.clone())
    assert torch.equal(outs[0], outs[1]) and torch.all(outs[0][Q * Q + Q:] == -3.0)
    xd = x.double()
    G2 = (xd.t() @ xd).cpu()
    zs = xd.sum(0).cpu()
    got = outs[0].double().cpu()
    assert torch.allclose(got[:Q * Q].view(Q, Q), G2, rtol=2e-5, atol=1e-3)
    assert torch.allclose(got[Q * Q:Q * Q + Q], zs, rtol=2e-5, atol=1e-3)


@pytest.mark.parametrize("case", [(2, 56, 256, 64), (3, 28, 512, 128), (1, 9, 256, 64), (7, 13, 512, 128), (2, 57, 256, 64)])
def test_two_source_data_gradient_with_bn_sums(case):
    """pfr_conv1x1_dgrad2_bn (csrc/pfr_sconv.hip EP 12): dx = [g | z]·wcatᵀ + bias over TWO row sources with the bias in the fp32
    accumulators, plus the BatchNorm-backward sums of the BN dx feeds (recomputed ReLU mask) — against fp64 torch on the same bf16
    operands (one bf16 rounding of the result), sums against the stored dx; ragged row counts; deterministic."""
    from pets_face_recognition_amd._hip import lib
    o = ops()
    N, H, C1, C2 = case
    Co = C2
    g = torch.Generator().manual_seed(H * C1 + C2)
    M = N * H * H
    G_ = (torch.randn(M, C1, generator=g) * (torch.rand(M, C1, generator=g) > 0.5)).bfloat16().to(DEV)
    Z = torch.relu(torch.randn(M, C2, generator=g)).bfloat16().to(DEV)
    wcat = (torch.randn(Co, C1 + C2, generator=g) / (C1 + C2) ** 0.5).bfloat16().to(DEV)
    bias = (0.1 * torch.randn(Co, generator=g)).to(DEV)
    # BN input away from the ReLU threshold (|scale*x + shift| >= 0.2): the recomputed mask must not depend on fma rounding
    bnx = ((0.5 + torch.rand(M, Co, generator=g)) * (torch.randint(0, 2, (M, Co), generator=g) * 2 - 1)).bfloat16().to(DEV)
    coef = torch.stack([0.1 * torch.randn(Co, generator=g), 1 + 0.1 * torch.rand(Co, generator=g),
                        1 + 0.1 * torch.rand(Co, generator=g), 0.2 * (torch.rand(Co, generator=g) - 0.5)]).to(DEV)   # mean, invstd, scale, shift
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    try:
        lib.pfr_set_tuning(b"sconv", 2)
        lib.pfr_set_tuning(b"bnb", 2)
        npart = lib.pfr_conv1x1_dgrad2_bn_parts(1, N, H, H, C1, C2, Co)
        assert npart > 0
        for _ in range(2):
            dx = torch.full((M + 1, Co), 9.0, dtype=torch.bfloat16, device=DEV)
            part = torch.full((npart + 1, 2, Co), 55.0, dtype=torch.float32, device=DEV)
            lib.pfr_conv1x1_dgrad2_bn(G_.data_ptr(), Z.data_ptr(), wcat.data_ptr(), bias.data_ptr(), dx.data_ptr(), 1, N, H, H, C1, C2, Co,
                                      bnx.data_ptr(), coef.data_ptr(), part.data_ptr(), st)
            torch.cuda.synchronize()
            outs.append((dx.clone(), part.clone()))
    finally:
        lib.pfr_set_tuning(b"sconv", 1)
        lib.pfr_set_tuning(b"bnb", 0)
    dx, part = outs[0]
    assert torch.equal(dx, outs[1][0]) and torch.equal(part, outs[1][1])
    assert torch.all(dx[M] == 9.0) and torch.all(part[npart] == 55.0)
    ref = torch.cat([G_, Z], 1).double() @ wcat.double().t() + bias.double()
    got = dx[:M].double()
    assert (got - ref).abs().max() <= 1e-2 * ref.abs().max()
    assert ((got - ref).norm() / ref.norm()).item() < 3e-3
    mask = (bnx.double() * coef[2].double() + coef[3].double()) > 0
    xh = (bnx.double() - coef[0].double()) * coef[1].double()
    s1 = (got * mask).sum(0)
    s2 = (got * mask * xh).sum(0)
    assert torch.allclose(part[:npart, 0].double().sum(0), s1, rtol=1e-4, atol=1e-2)
    assert torch.allclose(part[:npart, 1].double().sum(0), s2, rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("M,K,C", [(8192, 64, 256), (25088, 128, 512), (100357, 64, 256)])
def test_batchnorm_statistics_from_the_gram_matrix(M, K, C):
    """pfr_bn_stats_from_gram: mean / variance of x = Z·Wᵀ from Z's Gram matrix and column sums (no pass over x), through pfr_bn_finalize,
    against fp64 statistics of the exact product: mean to 1e-5 of the standard deviation, invstd to 1e-5 relative."""
    from pets_face_recognition_amd._hip import lib
    o = ops()
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(K, K, generator=g) / K ** 0.5
    Z = torch.relu(torch.randn(M, K, generator=g) @ A + 0.5).bfloat16().to(DEV)
    W = (torch.randn(C, K, generator=g) / K ** 0.5).bfloat16().to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(lib.pfr_gram_ws_floats(M, K), dtype=torch.float32, device=DEV)
    gram = torch.empty(K * K + K, dtype=torch.float32, device=DEV)
    lib.pfr_gram_colsum(Z.data_ptr(), 1, M, K, gram.data_ptr(), ws.data_ptr(), st)
    part = torch.empty(1, 2, C, dtype=torch.float32, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32).pin_memory()
    lib.pfr_bn_stats_from_gram(gram.data_ptr(), W.data_ptr(), 1, C, K, float(M), part.data_ptr(), 1e-5, flag.data_ptr(), st)
    coef = o.bn_finalize(part, M, M, None, None, 1e-5, 0.1, None, None)
    torch.cuda.synchronize()
    x = Z.double() @ W.double().t()
    mu, var = x.mean(0), x.var(0, unbiased=False)
    assert ((coef[0].double() - mu).abs() / var.sqrt()).max().item() < 1e-5
    r = (var + 1e-5).rsqrt()
    assert ((coef[1].double() - r).abs() / r).max().item() < 1e-5
    # statistics + finalize in one launch: the same to the last bit or two (pfr_bn_finalize's merge of the one row rounds once more)
    gm, bt = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    rm1, rv1 = torch.randn(C, device=DEV), torch.rand(C, device=DEV) + 0.5
    rm2, rv2 = rm1.clone(), rv1.clone()
    two = o.bn_finalize(part, M, M, gm, bt, 1e-5, 0.1, rm1, rv1)
    one = [torch.empty(C, dtype=torch.float32, device=DEV) for _ in range(4)]
    lib.pfr_bn_finalize_from_gram(gram.data_ptr(), W.data_ptr(), 1, C, K, float(M), gm.data_ptr(), bt.data_ptr(), 1e-5, 0.1, rm2.data_ptr(),
                                  rv2.data_ptr(), one[0].data_ptr(), one[1].data_ptr(), one[2].data_ptr(), one[3].data_ptr(), flag.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(flag[0]) == 0     # healthy statistics: the cancellation guard stays silent
    for a, b in zip(two, one):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)      # (shift = beta - mean * scale cancels: absolute tolerance)
    assert torch.allclose(rm1, rm2, rtol=1e-5, atol=1e-6) and torch.allclose(rv1, rv2, rtol=1e-5, atol=1e-6)


def test_gram_statistics_report_a_cancelled_variance():
    """var_c = W_c (G2/M - zbar zbar^T) W_c^T is an E[x^2] - E[x]^2 form.  A conv3 channel that is nearly constant (|mean| >> std)
    loses its variance to fp32 rounding; the kernel must SAY so (host-visible flag -> the engine falls back to the statistics pass), and
    must stay silent on ordinary channels.  Here channel 0 of x is 5 + 0.03 * {0, 1} (bf16-exact), mean^2 / var ~ 1e5."""
    from pets_face_recognition_amd._hip import lib
    M, K, C = 50176, 64, 256
    g = torch.Generator().manual_seed(11)
    Z = torch.relu(torch.randn(M, K, generator=g) + 0.5)
    Z[:, 0] = 5.0 + 0.03125 * (torch.rand(M, generator=g) < 0.5).float()
    Z = Z.bfloat16().to(DEV)
    W = (torch.randn(C, K, generator=g) / K ** 0.5)
    W[0] = 0.0
    W[0, 0] = 1.0                      # x[:, 0] = Z[:, 0]
    W = W.bfloat16().to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(lib.pfr_gram_ws_floats(M, K), dtype=torch.float32, device=DEV)
    gram = torch.empty(K * K + K, dtype=torch.float32, device=DEV)
    lib.pfr_gram_colsum(Z.data_ptr(), 1, M, K, gram.data_ptr(), ws.data_ptr(), st)
    part = torch.empty(1, 2, C, dtype=torch.float32, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32).pin_memory()
    lib.pfr_bn_stats_from_gram(gram.data_ptr(), W.data_ptr(), 1, C, K, float(M), part.data_ptr(), 1e-5, flag.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(flag[0]) == 1
    # the other channels are still right
    x = Z.double() @ W.double().t()
    var = x.var(0, unbiased=False)
    assert ((part[0, 1, 1:].double() / M - var[1:]).abs() / var[1:]).max().item() < 1e-4
    # the same data without the constant channel: silent
    W2 = W.clone()
    W2[0] = W2[1]
    flag[0] = 0
    lib.pfr_bn_stats_from_gram(gram.data_ptr(), W2.data_ptr(), 1, C, K, float(M), part.data_ptr(), 1e-5, flag.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(flag[0]) == 0


W9_CASES = [
    # N, H, W, C, Cout (3x3, stride 1, pad 1): row slots of 16 / 32 / 64 positions, ragged row counts, image rows that straddle stages
    (3, 14, 14, 64, 64), (2, 9, 13, 128, 64), (5, 8, 8, 64, 192), (2, 28, 28, 128, 128), (3, 17, 20, 64, 128), (2, 30, 30, 64, 64),
    (2, 56, 56, 64, 64), (1, 3, 40, 64, 64), (1, 2, 62, 64, 64), (33, 14, 14, 256, 256), (7, 28, 28, 64, 128),
    (9, 7, 7, 128, 64), (3, 5, 4, 64, 64), (40, 7, 7, 512, 512),     # 8-position row slots: a k-group spans two image rows
]


@pytest.mark.parametrize("case", W9_CASES)
def test_halo_staged_3x3_weight_gradient(case):
    """pfr_wgrad9.hip against torch's fp32 convolution weight gradient of the same bf16 inputs, and against the tile kernel (the
    same products in another fp32 summation order)"""
    from pets_face_recognition_amd._hip import lib
    o = ops()
    N, H, W, C, Cout = case
    g = torch.Generator().manual_seed(4321)
    x = torch.randn(N, C, H, W, generator=g).bfloat16().float()
    dy = torch.randn(N, Cout, H, W, generator=g).bfloat16().float()
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, C, 3, 3), dy.double(), stride=1, padding=1).permute(0, 2, 3, 1).float()
    xd, dyd = nhwc(x).to(DEV, torch.bfloat16), nhwc(dy).to(DEV, torch.bfloat16)
    try:
        lib.pfr_set_tuning(b"wgrad9", 2)
        a = o.conv2d_wgrad(xd, dyd, 3, 3, 1, 1)
        a2 = o.conv2d_wgrad(xd, dyd, 3, 3, 1, 1)
        lib.pfr_set_tuning(b"wgrad9", 0)
        b = o.conv2d_wgrad(xd, dyd, 3, 3, 1, 1)
    finally:
        lib.pfr_set_tuning(b"wgrad9", 1)
    torch.cuda.synchronize()
    assert torch.equal(a, a2)                                   # deterministic
    scale = ref.abs().max().item()
    assert (a.cpu() - ref).abs().max().item() < 1e-5 * scale, (a.cpu() - ref).abs().max().item() / scale
    assert (a - b).abs().max().item() < 1e-5 * scale


@pytest.mark.parametrize("mode", [0, 2])
def test_3x3_weight_gradient_of_a_channel_slice_of_dy(mode):
    """pfr_conv2d_wgrad with lddy > Cout (dy is a channel slice of a wider tensor) on the tile kernel and on pfr_wgrad9.hip: the same
    numbers as for a contiguous copy of the slice"""
    from pets_face_recognition_amd._hip import lib
    N, H, W, C, Cout, ld = 5, 14, 14, 64, 64, 192
    g = torch.Generator().manual_seed(8)
    x = torch.randn(N, H, W, C, generator=g).bfloat16().to(DEV)
    wide = torch.randn(N, H, W, ld, generator=g).bfloat16().to(DEV)
    dy_c = wide[..., :Cout].contiguous()
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    try:
        lib.pfr_set_tuning(b"wgrad9", mode)
        splits = lib.pfr_conv2d_wgrad_splits(N * H * W, Cout, 9 * C)
        ws = torch.empty(splits * Cout * 9 * C, dtype=torch.float32, device=DEV)
        for dy, lddy in ((dy_c, Cout), (wide, ld)):
            dw = torch.empty(Cout, 3, 3, C, dtype=torch.float32, device=DEV)
            lib.pfr_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), 1, N, H, W, C, Cout, 3, 3, 1, 1, H, W, lddy,
                                 0, 0, 0, 1.0, 0, st)
            outs.append(dw)
    finally:
        lib.pfr_set_tuning(b"wgrad9", 1)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2).double().cpu(), (Cout, C, 3, 3),
                                      dy_c.float().permute(0, 3, 1, 2).double().cpu(), stride=1, padding=1).permute(0, 2, 3, 1).float()
    assert (outs[0].cpu() - ref).abs().max().item() < 1e-5 * ref.abs().max().item()
