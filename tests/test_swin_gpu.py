"""Swin-T on the HIP path vs the reference-pinned torch restatement (models/swin.py on CPU == /root/reference/models/
swin.py, see tests/test_oracle_golden.py::test_swin_restatement_matches_reference_vectors) and the golden embeddings
generated from the reference itself (tests/golden/swin_t.npz)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shift", [0, 3])
def test_window_attention_kernel(dtype, shift):
    from pets_face_recognition_amd._hip import lib, dtype_id
    import pets_face_recognition_amd.models.swin as S
    g = torch.Generator().manual_seed(4)
    B, H, W, heads, hd, w = 2, 14, 14, 3, 32, 7
    C = heads * hd
    att = S.WindowAttention(C, heads, hd, shift > 0, w, True)
    qkv = torch.randn(B, H, W, 3 * C, generator=g) * 0.7
    if dtype == torch.bfloat16:
        qkv = qkv.bfloat16().float()
    qkv.requires_grad_(True)
    # torch reference of the core (everything of WindowAttention.forward between to_qkv and to_out)
    x = qkv
    if shift:
        x = torch.roll(x, (-shift, -shift), (1, 2))
    gh, gw = H // w, W // w
    q, k, v = (att._windows(t, B, gh, gw) for t in x.chunk(3, dim=-1))
    dots = torch.matmul(q, k.transpose(-1, -2)) * att.scale
    ri = att.relative_indices
    dots = dots + att.pos_embedding[ri[:, :, 0], ri[:, :, 1]]
    if shift:
        dots[:, :, -gw:] += att.upper_lower_mask
        dots[:, :, gw - 1::gw] += att.left_right_mask
    out = torch.matmul(dots.softmax(dim=-1), v)
    out = out.view(B, heads, gh, gw, w, w, -1).permute(0, 2, 4, 3, 5, 1, 6).reshape(B, H, W, -1)
    if shift:
        out = torch.roll(out, (shift, shift), (1, 2))
    dout = torch.randn(out.shape, generator=g)
    if dtype == torch.bfloat16:
        dout = dout.bfloat16().float()
    out.backward(dout)
    # HIP
    st = torch.cuda.current_stream().cuda_stream
    qd = qkv.detach().to(DEV, dtype).contiguous()
    pos = att.pos_embedding.detach().to(DEV).contiguous()
    od = torch.empty(B, H, W, C, dtype=dtype, device=DEV)
    tab = torch.empty(lib.pfr_window_bias_table_floats(w), dtype=torch.float32, device=DEV)
    lib.pfr_window_bias_table(pos.data_ptr(), tab.data_ptr(), w, shift, st)
    pos = tab
    lib.pfr_window_attn_fwd(qd.data_ptr(), pos.data_ptr(), od.data_ptr(), dtype_id(dtype), B, H, W, heads, hd, w, shift,
                            float(att.scale), st)
    dq = torch.empty_like(qd)
    nblk = B * gh * gw * heads
    dpart = torch.empty(nblk, 169, dtype=torch.float32, device=DEV)
    dod = dout.to(DEV, dtype).contiguous()
    lib.pfr_window_attn_bwd(qd.data_ptr(), pos.data_ptr(), dod.data_ptr(), dq.data_ptr(),
                            dpart.data_ptr(), dtype_id(dtype), B, H, W, heads, hd, w, shift, float(att.scale), st)
    torch.cuda.synchronize()
    t = 2e-2 if dtype == torch.bfloat16 else 2e-5
    assert rel(od, out.detach()) < t
    assert rel(dq, qkv.grad) < t
    assert rel(dpart.sum(0), att.pos_embedding.grad.flatten()) < t


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_gelu_kernels(dtype):
    from pets_face_recognition_amd._hip import lib, dtype_id
    g = torch.Generator().manual_seed(6)
    rows, C = 1000, 96
    x = torch.randn(rows, C, generator=g) * 2 + 0.3
    res = torch.randn(rows, C, generator=g)
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    dy = torch.randn(rows, C, generator=g)
    if dtype == torch.bfloat16:
        x, res, dy = x.bfloat16().float(), res.bfloat16().float(), dy.bfloat16().float()
    xr = x.clone().requires_grad_(True); gr = gam.clone().requires_grad_(True); br = bet.clone().requires_grad_(True)
    y = F.layer_norm(xr, (C,), gr, br, 1e-5)
    y.backward(dy)
    st = torch.cuda.current_stream().cuda_stream
    did = dtype_id(dtype)
    xd = x.to(DEV, dtype); yd = torch.empty_like(xd)
    mu = torch.empty(rows, device=DEV); rs = torch.empty(rows, device=DEV)
    gd, bd, dyd, resd = gam.to(DEV), bet.to(DEV), dy.to(DEV, dtype), res.to(DEV, dtype)   # keep the device buffers alive
    lib.pfr_layernorm_fwd(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), yd.data_ptr(), mu.data_ptr(), rs.data_ptr(),
                          did, rows, C, 1e-5, st)
    nb = lib.pfr_layernorm_bwd_blocks(rows)
    part = torch.empty(2, nb, C, device=DEV); dx = torch.empty_like(xd)
    lib.pfr_layernorm_bwd(dyd.data_ptr(), xd.data_ptr(), mu.data_ptr(), rs.data_ptr(), gd.data_ptr(),
                          resd.data_ptr(), dx.data_ptr(), part.data_ptr(), did, rows, C, st)
    torch.cuda.synchronize()
    t = 2e-2 if dtype == torch.bfloat16 else 2e-5
    assert rel(yd, y.detach()) < t
    assert rel(dx, xr.grad + res) < t
    assert rel(part[0].sum(0), gr.grad) < t and rel(part[1].sum(0), br.grad) < t
    # GELU
    h = torch.randn(rows, 4 * C, generator=g) * 1.5
    if dtype == torch.bfloat16:
        h = h.bfloat16().float()
    hr = h.clone().requires_grad_(True)
    gy = F.gelu(hr)
    dg = torch.randn(gy.shape, generator=g)
    gy.backward(dg)
    hd_ = h.to(DEV, dtype); o = torch.empty_like(hd_); dh = torch.empty_like(hd_); dgd = dg.to(DEV, dtype)
    lib.pfr_gelu_fwd(hd_.data_ptr(), o.data_ptr(), did, h.numel(), st)
    lib.pfr_gelu_bwd(hd_.data_ptr(), dgd.data_ptr(), dh.data_ptr(), did, h.numel(), st)
    torch.cuda.synchronize()
    assert rel(o, gy.detach()) < t and rel(dh, hr.grad) < t


def test_swin_t_embeddings_match_reference_golden():
    """fp32 HIP path reproduces the embeddings the REFERENCE's swin_t produced for the same seed / input"""
    import pets_face_recognition_amd.models as M
    G = np.load(os.path.join(GOLD, "swin_t.npz"))
    torch.manual_seed(int(G["seed"]))
    m = M.swin_t(num_classes=512, compute_dtype=torch.float32).to(DEV).eval()
    x = torch.rand(2, 3, 224, 224, generator=torch.Generator().manual_seed(int(G["x_seed"])))
    with torch.no_grad():
        y = m(x.to(DEV))
    torch.cuda.synchronize()
    assert rel(y, torch.tensor(G["emb"])) < 1e-3


@pytest.mark.parametrize("dtype,tol_e,tol_g", [(torch.float32, 1e-3, 5e-3), (torch.bfloat16, 5e-2, 1.5e-1)])
def test_swin_fwd_bwd_vs_torch_restatement(dtype, tol_e, tol_g):
    import pets_face_recognition_amd.models as M
    torch.manual_seed(21)
    ref = M.swin_t(num_classes=512)                     # CPU torch path == reference (pinned by golden test)
    hip = M.swin_t(num_classes=512, compute_dtype=dtype)
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(DEV)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(3, 3, 224, 224, generator=g)
    demb = torch.randn(3, 512, generator=g) * 0.1
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    e_ref = ref(x)
    e_ref.backward(demb)
    e = hip(x.to(DEV))
    e.backward(demb.to(DEV))
    torch.cuda.synchronize()
    assert rel(e, e_ref.detach()) < tol_e
    rp = dict(ref.named_parameters())
    worst = ("", 0.0)
    fh, fr = [], []
    for n, p in hip.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, n
        r = rel(p.grad, rp[n].grad)
        fh.append(p.grad.double().cpu().flatten()); fr.append(rp[n].grad.double().flatten())
        if r > worst[1]:
            worst = (n, r)
    cos = F.cosine_similarity(torch.cat(fh), torch.cat(fr), dim=0).item()
    assert cos > (0.99999 if dtype == torch.float32 else 0.99), cos
    assert worst[1] < tol_g, worst


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_with_fused_gelu_epilogues(dtype):
    """pfr_gemm_act: act 2 (y2 = x·Wᵀ + b, y = gelu(y2)) and act 3 (y = (x·Wᵀ) ∘ gelu'(y2)) vs torch (exact erf GELU)."""
    from pets_face_recognition_amd._hip import lib, dtype_id
    g = torch.Generator().manual_seed(8)
    M, K, N = 777, 96, 384          # ragged M (tile tail)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    st = torch.cuda.current_stream().cuda_stream
    xd, wd, bd = x.to(DEV, dtype), w.to(DEV, dtype), b.to(DEV)
    y = torch.empty(M, N, dtype=dtype, device=DEV); y2 = torch.empty_like(y)
    lib.pfr_gemm_act(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), dtype_id(dtype), M, K, N, bd.data_ptr(), 2, y2.data_ptr(), st)
    torch.cuda.synchronize()
    z = x @ w.t() + b
    t = 1e-2 if dtype == torch.bfloat16 else 2e-5
    assert rel(y2, z) < t
    assert rel(y, F.gelu(y2.float().cpu())) < t          # GELU of the STORED pre-activation
    # act 3: data gradient of a Linear(N -> K) joined with GELU backward at its input (pre-activation pre[M][K])
    pre = torch.randn(M, K, generator=g)
    dy = torch.randn(M, N, generator=g)
    if dtype == torch.bfloat16:
        pre, dy = pre.bfloat16().float(), dy.bfloat16().float()
    wt = w.t().contiguous()                                # [K][N]: rows = outputs of the gradient GEMM
    out = torch.empty(M, K, dtype=dtype, device=DEV)
    pd = pre.to(DEV, dtype)
    lib.pfr_gemm_act(dy.to(DEV, dtype).data_ptr(), wt.to(DEV, dtype).data_ptr(), out.data_ptr(), dtype_id(dtype), M, N, K, 0, 3,
                     pd.data_ptr(), st)
    torch.cuda.synchronize()
    pr = pre.clone().requires_grad_(True)
    F.gelu(pr).backward(dy @ w)
    assert rel(out, pr.grad) < t


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C", [(64, 96), (1000, 96), (401408, 96), (100, 192), (25088, 384), (2, 768), (3, 768), (6272, 768), (37, 1024), (5000, 32)])
@pytest.mark.parametrize("use_res", [1, 0])
def test_layernorm_shapes_vs_torch(dtype, rows, C, use_res):
    """pfr_layernorm_fwd / pfr_layernorm_bwd_dxsum against torch over the batch / lane-group geometries of the kernels: one to four
    16-byte chunks per lane, rows that do not fill the last batch of a workgroup, lanes past the row's last chunk (C = 96, 192, 384: the
    range-checked buffer addressing reads zeros there and drops the stores), with and without the residual gradient
    (reference: nn.LayerNorm of /root/reference/models/swin.py:28-36 PreNorm)"""
    from pets_face_recognition_amd._hip import lib, dtype_id
    dev = "cuda:0"
    st = torch.cuda.current_stream().cuda_stream
    did = dtype_id(dtype)
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.3).to(dev, dtype)
    dy = torch.randn(rows, C, generator=g).to(dev, dtype)
    res = torch.randn(rows, C, generator=g).to(dev, dtype)
    gam = (torch.rand(C, generator=g) + 0.5).to(dev)
    bet = (torch.randn(C, generator=g) * 0.1).to(dev)
    xr = x.float().clone().requires_grad_(True); gr = gam.clone().requires_grad_(True); br = bet.clone().requires_grad_(True)
    y = F.layer_norm(xr, (C,), gr, br, 1e-5)
    y.backward(dy.float())
    yd = torch.empty_like(x); mu = torch.empty(rows, device=dev); rs = torch.empty(rows, device=dev)
    lib.pfr_layernorm_fwd(x.data_ptr(), gam.data_ptr(), bet.data_ptr(), yd.data_ptr(), mu.data_ptr(), rs.data_ptr(), did, rows, C, 1e-5, st)
    nb = lib.pfr_layernorm_bwd_blocks(rows)
    part = torch.empty(2, nb, C, device=dev); dx = torch.empty_like(x)
    ok = lib.pfr_layernorm_bwd_dxsum_ok(did, C)
    dsum = torch.empty(nb, C, device=dev) if ok else None
    lib.pfr_layernorm_bwd_dxsum(dy.data_ptr(), x.data_ptr(), mu.data_ptr(), rs.data_ptr(), gam.data_ptr(), res.data_ptr() if use_res else 0,
                                dx.data_ptr(), part.data_ptr(), dsum.data_ptr() if ok else 0, did, rows, C, st)
    torch.cuda.synchronize()
    t = 3e-2 if dtype == torch.bfloat16 else 1e-4
    assert rel(yd, y.detach()) < t
    assert rel(mu, x.float().mean(1)) < 1e-4 and rel(rs, (x.float().var(1, unbiased=False) + 1e-5).rsqrt()) < 1e-4
    assert rel(dx, xr.grad + (res.float() if use_res else 0)) < t
    assert rel(part[0].sum(0), gr.grad) < t and rel(part[1].sum(0), br.grad) < t
    if ok:
        assert rel(dsum.sum(0), dx.float().sum(0)) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C", [(300, 96), (4099, 192), (257, 768)])
def test_layernorm_bwd_dxsum_partials(dtype, rows, C):
    """pfr_layernorm_bwd_dxsum: dx and the dγ/dβ partials are those of pfr_layernorm_bwd, and the extra partial rows sum to the
    column sums of dx as stored (the bias gradient of the layer in front of the LayerNorm)"""
    from pets_face_recognition_amd._hip import lib, dtype_id
    dev = "cuda:0"
    did = dtype_id(dtype)
    if not lib.pfr_layernorm_bwd_dxsum_ok(did, C):
        pytest.skip("generic LayerNorm kernel for this channel count")
    g = torch.Generator().manual_seed(2)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(rows, C, generator=g).to(dev, dtype)
    dy = torch.randn(rows, C, generator=g).to(dev, dtype)
    dres = torch.randn(rows, C, generator=g).to(dev, dtype)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    beta = torch.zeros(C, device=dev)
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    lib.pfr_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), did, rows, C, 1e-5, st)
    nb = lib.pfr_layernorm_bwd_blocks(rows)
    dx_a = torch.empty_like(x); part_a = torch.zeros(2, nb, C, device=dev)
    lib.pfr_layernorm_bwd(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dres.data_ptr(), dx_a.data_ptr(),
                          part_a.data_ptr(), did, rows, C, st)
    dx_b = torch.empty_like(x); part_b = torch.zeros(2, nb, C, device=dev); dsum = torch.full((nb, C), float("nan"), device=dev)
    lib.pfr_layernorm_bwd_dxsum(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dres.data_ptr(),
                                dx_b.data_ptr(), part_b.data_ptr(), dsum.data_ptr(), did, rows, C, st)
    torch.cuda.synchronize()
    assert torch.equal(dx_a, dx_b) and torch.equal(part_a, part_b)
    ref = dx_b.double().sum(0)
    got = dsum.double().sum(0)
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-4 * float(dx_b.float().abs().max()) * rows ** 0.5)


def test_swin_backward_under_the_persistent_gemm_and_plan_invalidation():
    """(a) the Swin train step under pfr_set_tuning("igemm_p", 2) (persistent GEMM kernel wherever eligible:
    other statistics-partial granularity) gives the gradients of the default kernels — the persistent kernel is bit-identical, so
    the step is too; (b) flipping the knob AFTER the plans were built makes the engine rebuild them (pfr_tuning_epoch) instead of
    replaying launches whose baked tile heights no longer describe the kernel that runs."""
    import pets_face_recognition_amd.models as M
    from pets_face_recognition_amd._hip import lib
    torch.manual_seed(3)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(4, 3, 224, 224, generator=g).to(DEV)
    demb = (torch.randn(4, 512, generator=g) * 0.1).to(DEV)
    m = M.swin_t(num_classes=512, compute_dtype=torch.bfloat16).to(DEV).train()

    def step():
        for p in m.parameters():
            p.grad = None
        e = m(x)
        e.backward(demb)
        torch.cuda.synchronize()
        return e.detach().clone(), torch.cat([p.grad.flatten() for p in m.parameters() if p.requires_grad]).clone()

    try:
        e0, g0 = step()
        plans0 = {id(p) for p in m.hip_engine().plans.values()}
        ep0 = lib.pfr_tuning_epoch()
        lib.pfr_set_tuning(b"igemm_p", 2)
        assert lib.pfr_tuning_epoch() == ep0 + 1
        lib.pfr_set_tuning(b"igemm_p", 2)            # no change: no new epoch
        assert lib.pfr_tuning_epoch() == ep0 + 1
        e1, g1 = step()
        assert not (plans0 & {id(p) for p in m.hip_engine().plans.values()}), "plans built under the old knobs were replayed"
        assert torch.equal(e0, e1) and torch.isfinite(g1).all()
        assert torch.equal(g0, g1)
    finally:
        lib.pfr_set_tuning(b"igemm_p", 1)


SLIN_CASES = [
    # rows, K, N, bias, residual: Swin stage-1 / stage-2 Linear shapes (scaled rows), ragged row counts, every panel width (64 / 96 / 192)
    (8192, 96, 96, True, True), (8192 + 17, 96, 288, True, False), (6000, 288, 96, False, False), (8192, 96, 384, True, False),
    (5000, 384, 96, True, True), (4096 + 5, 192, 192, True, True), (4100, 192, 576, True, False), (4096, 576, 192, False, False),
    (4096, 192, 768, True, False), (4097, 768, 192, True, True), (4096, 384, 384, True, True), (31 * 133, 96, 64, True, False),
]


@pytest.mark.gpu
@pytest.mark.parametrize("rows,K,N,has_bias,has_res", SLIN_CASES)
def test_streaming_linear_kernel_bit_identical(rows, K, N, has_bias, has_res):
    """csrc/pfr_slin.hip (round 5): the weight-stationary streaming Linear kernel for K = 96 j must give the tile kernel's bits — same k
    order inside and across the MFMAs, same epilogue arithmetic (round to bf16, add residual and bias in fp32, round) — for the plain,
    bias, residual forms, the fused GELU (act 2: pre-activation AND activation) and GELU backward (act 3) of pfr_gemm_act."""
    from pets_face_recognition_amd._hip import lib, ops
    g = torch.Generator().manual_seed(rows + K + N)
    x = torch.randn(rows, 1, 1, K, generator=g).bfloat16().to(DEV)
    w = (torch.randn(N, 1, 1, K, generator=g) / K ** 0.5).bfloat16().to(DEV)
    bias = torch.randn(N, generator=g).to(DEV) if has_bias else None
    res = torch.randn(rows, 1, 1, N, generator=g).bfloat16().to(DEV) if has_res else None
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    try:
        for mode in (0, 2):
            lib.pfr_set_tuning(b"slin", mode)
            y, _ = ops.conv2d_fwd(x, w, bias=bias, residual=res)
            r = [y.clone()]
            if not has_res:
                # fused GELU forward: y2 = pre-activation, y = gelu(y2); then GELU backward on a data gradient of the same shape
                bz = bias if bias is not None else torch.zeros(N, device=DEV)
                h1 = torch.empty(rows, N, dtype=torch.bfloat16, device=DEV)
                h2 = torch.empty(rows, N, dtype=torch.bfloat16, device=DEV)
                lib.pfr_gemm_act(x.data_ptr(), w.data_ptr(), h2.data_ptr(), 1, rows, K, N, bz.data_ptr(), 2, h1.data_ptr(), st)
                dz = torch.empty(rows, N, dtype=torch.bfloat16, device=DEV)
                lib.pfr_gemm_act(x.data_ptr(), w.data_ptr(), dz.data_ptr(), 1, rows, K, N, 0, 3, h1.data_ptr(), st)
                r += [h1, h2, dz]
            torch.cuda.synchronize()
            out[mode] = r
    finally:
        lib.pfr_set_tuning(b"slin", 1)
    for a, b_ in zip(out[0], out[2]):
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a, b_)
    # and against fp32 torch (the tile kernel is pinned to it elsewhere; this guards the test itself against comparing two no-ops)
    ref = x.view(rows, K).float() @ w.view(N, K).float().t()
    if res is not None:
        ref = ref + res.view(rows, N).float()
    if bias is not None:
        ref = ref + bias
    assert rel(out[2][0].view(rows, N), ref) < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("rows,K,N", [(8192 + 9, 96, 384), (4096, 96, 96), (5000, 192, 768), (4100, 96, 288)])
def test_streaming_linear_gelu_backward_with_column_sums(rows, K, N):
    """pfr_gemm_act_colsums (round 5): GELU backward on the data gradient + plain column sums of the stored output, one partial row per row
    range of the streaming Linear kernel — the bias gradient of the Linear in front without a pass over the gradient.  The output must be
    the tile kernel's bits (pfr_gemm_act, act 3); the partial rows must add up to the fp32 column sum of the STORED bf16 output."""
    from pets_face_recognition_amd._hip import lib
    g = torch.Generator().manual_seed(rows + N)
    x = torch.randn(rows, K, generator=g).bfloat16().to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().to(DEV)
    z = torch.randn(rows, N, generator=g).bfloat16().to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    try:
        lib.pfr_set_tuning(b"slin", 0)
        assert lib.pfr_gemm_act_colsum_parts(rows, K, N, 1) == 0
        ref = torch.empty(rows, N, dtype=torch.bfloat16, device=DEV)
        lib.pfr_gemm_act(x.data_ptr(), w.data_ptr(), ref.data_ptr(), 1, rows, K, N, 0, 3, z.data_ptr(), st)
        lib.pfr_set_tuning(b"slin", 2)
        parts = lib.pfr_gemm_act_colsum_parts(rows, K, N, 1)
        assert parts > 0
        y = torch.empty(rows, N, dtype=torch.bfloat16, device=DEV)
        sums = torch.full((parts, N), float("nan"), dtype=torch.float32, device=DEV)
        lib.pfr_gemm_act_colsums(x.data_ptr(), w.data_ptr(), y.data_ptr(), 1, rows, K, N, z.data_ptr(), sums.data_ptr(), st)
        torch.cuda.synchronize()
    finally:
        lib.pfr_set_tuning(b"slin", 1)
    assert torch.equal(y, ref)
    assert torch.isfinite(sums).all()                      # every partial row was written
    want = y.float().sum(0)
    got = sums.sum(0)
    assert ((got - want).abs() / (want.abs() + y.float().abs().sum(0) * 1e-5 + 1e-6)).max().item() < 1e-3
    # through the batched final merge the engine uses (mt = 0: plain partial rows)
    import struct
    out = torch.zeros(N, dtype=torch.float32, device=DEV)
    raw = struct.pack("<QQiiiiii", sums.data_ptr(), out.data_ptr(), parts, N, 0, 0, 0, 0)
    tab = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(DEV)
    lib.pfr_colsum_final_batch(tab.data_ptr(), 1, N, st)
    torch.cuda.synchronize()
    assert torch.allclose(out, got, rtol=1e-5, atol=1e-4)


@pytest.mark.gpu
def test_bias_tables_of_all_blocks_in_one_launch_equal_the_per_block_launches():
    """pfr_window_bias_table_batch (one launch per forward pass) writes what pfr_window_bias_table writes per block: regular and shifted
    windows, both copies of the table (models/swin.py:65-70,93-95 relative position bias, :49-62,86-90 shifted-window masks)"""
    import struct
    from pets_face_recognition_amd._hip import lib
    dev = "cuda:0"
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(11)
    w = 7
    n = lib.pfr_window_bias_table_floats(w)
    blocks = [(torch.randn(2 * w - 1, 2 * w - 1, generator=g).to(dev), s) for s in (0, 3, 0, 3, 0)]
    one = [torch.full((n,), float("nan"), device=dev) for _ in blocks]
    allb = [torch.full((n,), float("nan"), device=dev) for _ in blocks]
    for (pos, s), t in zip(blocks, one):
        lib.pfr_window_bias_table(pos.data_ptr(), t.data_ptr(), w, s, st)
    raw = b"".join(struct.pack("<QQii", pos.data_ptr(), t.data_ptr(), w, s) for (pos, s), t in zip(blocks, allb))
    descs = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    lib.pfr_window_bias_table_batch(descs.data_ptr(), len(blocks), st)
    torch.cuda.synchronize()
    for a, b in zip(one, allb):
        assert torch.equal(a, b) and not torch.isnan(a).any()


@pytest.mark.gpu
@pytest.mark.parametrize("rows,C", [(1000, 96), (100352, 192), (777, 384)])
def test_layernorm_forward_rows_in_flight_knob_is_bit_identical(rows, C):
    """the `ln_rb` knob (rows in flight per lane group of pfr_layernorm_fwd: 4 | 6 | 8) only changes which rows a workgroup takes"""
    from pets_face_recognition_amd._hip import lib, dtype_id
    dev = "cuda:0"
    st = torch.cuda.current_stream().cuda_stream
    did = dtype_id(torch.bfloat16)
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, C, generator=g) * 1.5 - 0.2).to(dev, torch.bfloat16)
    gam = (torch.rand(C, generator=g) + 0.5).to(dev); bet = (torch.randn(C, generator=g) * 0.1).to(dev)
    outs = []
    try:
        for rb in (4, 6, 8):
            assert lib.pfr_set_tuning(b"ln_rb", rb) == 0
            y = torch.full_like(x, float("nan")); mu = torch.full((rows,), float("nan"), device=dev); rs = torch.full((rows,), float("nan"), device=dev)
            lib.pfr_layernorm_fwd(x.data_ptr(), gam.data_ptr(), bet.data_ptr(), y.data_ptr(), mu.data_ptr(), rs.data_ptr(), did, rows, C, 1e-5, st)
            torch.cuda.synchronize()
            outs.append((y, mu, rs))
    finally:
        lib.pfr_set_tuning(b"ln_rb", 4)
    for y, mu, rs in outs[1:]:
        assert torch.equal(y, outs[0][0]) and torch.equal(mu, outs[0][1]) and torch.equal(rs, outs[0][2])
    assert not torch.isnan(outs[0][0].float()).any()
