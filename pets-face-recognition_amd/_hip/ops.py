"""Tensor-level wrappers over the C-ABI (include/pfr_hip.h).  Tensors are torch CUDA tensors used purely as
device-memory handles: `.data_ptr()` goes straight into libpfr_hip.so, launches go on torch's current stream.
Nothing here computes with torch."""
import torch

from .lib import lib, dtype_id, PFR_F32, PFR_BF16, PfrError


def _p(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name):
    if t is not None:
        if not t.is_cuda:
            raise PfrError(f"{name}: expected a CUDA (HIP) tensor — the HIP path has no CPU fallback")
        if not t.is_contiguous():
            raise PfrError(f"{name}: tensor must be contiguous")


def kpack(dtype):
    return 8 if dtype == torch.bfloat16 else 4


# ------------------------------------------------------------------------------------------------ conv / linear
def conv_out_hw(H, W, R, S, stride, pad):
    return (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1


def conv2d_fwd(x, w, stride=1, pad=0, idil_log2=0, out_hw=None, out=None, out_dtype=None, bias=None, accumulate=False,
               out_relu=False, pro=None, stats=False, stats_buf=None, residual=None):
    """x [N,H,W,C] NHWC, w [Cout,R,S,C].  Returns (y [N,OH,OW,Cout], stats_part or None)."""
    _chk(x, "x"); _chk(w, "w")
    N, H, W, C = x.shape
    Cout, R, S, Cw = w.shape
    assert Cw == C, (Cw, C)
    if out_hw is None:
        OH, OW = conv_out_hw(H, W, R, S, stride, pad)
    else:
        OH, OW = out_hw
    odt = out_dtype or x.dtype
    if out is None:
        out = torch.empty((N, OH, OW, Cout), dtype=odt, device=x.device)
    ldy = out.shape[-1]
    M = N * OH * OW
    part = None
    if stats:
        mt = lib.pfr_conv2d_mtile(N, H, W, C, Cout, R, S, stride, pad, OH, OW, dtype_id(x.dtype), dtype_id(out.dtype), int(pro is not None))
        nt = (M + mt - 1) // mt
        conv2d_fwd.last_mt = mt     # (rows per statistics partial of this launch: what pfr_bn_finalize needs as rows_per_part)
        part = stats_buf if stats_buf is not None else torch.empty((nt, 2, Cout), dtype=torch.float32, device=x.device)
        assert part.numel() >= nt * 2 * Cout
    ps = psh = None
    prelu = 0
    if pro is not None:
        ps, psh, prelu = pro
    lib.pfr_conv2d_fwd(_p(x), _p(w), _p(out), dtype_id(x.dtype), dtype_id(out.dtype), N, H, W, C, Cout, R, S, stride, pad,
                       idil_log2, OH, OW, ldy, _p(bias), _p(residual), int(accumulate), int(out_relu), _p(ps), _p(psh), int(prelu),
                       _p(part), _stream())
    return out, part


def conv2d_dgrad(dy, wt, in_hw, stride, pad, R, S, out=None, accumulate=False, out_dtype=None):
    """dy [N,OH,OW,Cout]; wt [Cin,R,S,Cout] = dgrad layout of the weights (weight_dgrad_layout).  → dx [N,H,W,Cin]"""
    H, W = in_hw
    log2 = {1: 0, 2: 1, 4: 2}[stride]
    y, _ = conv2d_fwd(dy, wt, stride=1, pad=R - 1 - pad, idil_log2=log2, out_hw=(H, W), out=out, accumulate=accumulate,
                      out_dtype=out_dtype)
    return y


def _conv2d_dgrad_geom_check(dy, in_hw, stride, pad, R):
    # the dilated-input gather computes ih = (h - (R-1-pad) + r') which must equal h + pad - r with r' = R-1-r
    return True


def weight_dgrad_layout(w, out=None):
    O, R, S, I = w.shape
    if out is None:
        out = torch.empty((I, R, S, O), dtype=w.dtype, device=w.device)
    lib.pfr_weight_dgrad_layout(_p(w), _p(out), dtype_id(w.dtype), O, R, S, I, _stream())
    return out


def conv2d_wgrad(x, dy, R, S, stride, pad, out=None, pro=None, scale=1.0, accumulate=False, workspace=None):
    """x [N,H,W,C], dy [N,OH,OW,Cout] → dw fp32 [Cout,R,S,C]"""
    _chk(x, "x"); _chk(dy, "dy")
    N, H, W, C = x.shape
    _, OH, OW, Cout = dy.shape
    if out is None:
        out = torch.empty((Cout, R, S, C), dtype=torch.float32, device=x.device)
    KK = R * S * C
    splits = lib.pfr_conv2d_wgrad_splits(N * OH * OW, Cout, KK)
    need = splits * Cout * KK
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.float32, device=x.device)
    ps = psh = None
    prelu = 0
    if pro is not None:
        ps, psh, prelu = pro
    lib.pfr_conv2d_wgrad(_p(x), _p(dy), _p(out), _p(workspace), dtype_id(x.dtype), N, H, W, C, Cout, R, S, stride, pad, OH,
                         OW, Cout, _p(ps), _p(psh), int(prelu), float(scale), int(accumulate), _stream())
    return out


# ------------------------------------------------------------------------------------------------ layout
def nchw_to_nhwc(x, dtype, cpad):
    N, C, H, W = x.shape
    assert x.dtype == torch.float32
    _chk(x, "x")
    y = torch.empty((N, H, W, cpad), dtype=dtype, device=x.device)
    lib.pfr_nchw_to_nhwc(_p(x), _p(y), dtype_id(dtype), N, C, H, W, cpad, _stream())
    return y


def cast(x, dtype, out=None):
    if out is None:
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
    lib.pfr_cast(_p(x), dtype_id(x.dtype), _p(out), dtype_id(dtype), x.numel(), _stream())
    return out


def add(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    lib.pfr_add(_p(a), _p(b), _p(out), dtype_id(a.dtype), a.numel(), _stream())
    return out


def colsum(x, out=None, accumulate=False):
    rows, C = x.shape
    if out is None:
        out = torch.empty(C, dtype=torch.float32, device=x.device)
    nws = lib.pfr_colsum_ws_floats(rows, C)
    ws = torch.empty(nws, dtype=torch.float32, device=x.device) if nws else None
    lib.pfr_colsum(_p(x), dtype_id(x.dtype), rows, C, _p(out), int(accumulate), _p(ws), _stream())
    return out


# ------------------------------------------------------------------------------------------------ batch norm
def bn_stats(x):
    C = x.shape[-1]
    rows = x.numel() // C
    nb = lib.pfr_colreduce_blocks(C, dtype_id(x.dtype), rows)
    part = torch.empty((nb, 2, C), dtype=torch.float32, device=x.device)
    lib.pfr_bn_stats(_p(x), dtype_id(x.dtype), rows, C, _p(part), _stream())
    return part, lib.pfr_bn_stats_rows_per_part(C, dtype_id(x.dtype), rows)


def bn_finalize(part, rows_per_part, count, gamma, beta, eps, momentum, running_mean, running_var, out=None):
    """→ (mean, invstd, scale, shift) rows of a [4,C] fp32 tensor"""
    C = part.shape[-1]
    nparts = part.numel() // (2 * C)
    if out is None:
        out = torch.empty((4, C), dtype=torch.float32, device=part.device)
    nws = lib.pfr_bn_finalize_ws_floats(nparts, C)
    ws = torch.empty(nws, dtype=torch.float32, device=part.device) if nws else None
    lib.pfr_bn_finalize(_p(part), nparts, int(rows_per_part), C, float(count), _p(gamma), _p(beta), float(eps), float(momentum),
                        _p(running_mean), _p(running_var), _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), _p(ws), _stream())
    return out


def bn_eval_coeff(gamma, beta, running_mean, running_var, eps, out=None):
    C = running_mean.numel()
    if out is None:
        out = torch.empty((2, C), dtype=torch.float32, device=running_mean.device)
    lib.pfr_bn_eval_coeff(C, _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(eps), _p(out[0]), _p(out[1]), _stream())
    return out


def bn_act(x1, a1, b1, x2=None, a2=None, b2=None, relu=True, out=None, want_mask=False):
    """-> out, or (out, mask) with the ReLU bit mask ([rows][C / KPACK] bytes) that bn_bwd(mask_mode=3) consumes."""
    C = x1.shape[-1]
    rows = x1.numel() // C
    if out is None:
        out = torch.empty_like(x1)
    if want_mask:
        kp = 8 if x1.dtype == torch.bfloat16 else 4
        mask = torch.empty((rows, C // kp), dtype=torch.uint8, device=x1.device)
        lib.pfr_bn_act_mask(_p(x1), _p(a1), _p(b1), _p(x2), _p(a2), _p(b2), _p(out), _p(mask), dtype_id(x1.dtype), rows, C,
                            int(relu), _stream())
        return out, mask
    lib.pfr_bn_act(_p(x1), _p(a1), _p(b1), _p(x2), _p(a2), _p(b2), _p(out), dtype_id(x1.dtype), rows, C, int(relu), _stream())
    return out


def bn_bwd(dout, x, mean, invstd, gamma, count, mask_mode=0, out_act=None, scale=None, shift=None, dgamma=None,
           dbeta=None, want_gres=False, dx=None, gres=None, accumulate_param_grads=False):
    """Full BN(+ReLU mask) backward.  Returns (dx, gres or None, dgamma, dbeta)."""
    C = x.shape[-1]
    rows = x.numel() // C
    did = dtype_id(x.dtype)
    nb = lib.pfr_colreduce_blocks(C, did, rows)
    part = torch.empty((nb, 2, C), dtype=torch.float32, device=x.device)
    lib.pfr_bn_bwd_reduce(_p(dout), _p(out_act), _p(x), _p(mean), _p(invstd), _p(scale), _p(shift), mask_mode, did, rows, C,
                          _p(part), _stream())
    coef = torch.empty((3, C), dtype=torch.float32, device=x.device)
    if dgamma is None:
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
    if dbeta is None:
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    lib.pfr_bn_bwd_finalize(_p(part), nb, C, float(count), _p(gamma), _p(mean), _p(invstd), _p(dgamma), _p(dbeta), _p(coef),
                            int(accumulate_param_grads), _stream())
    if dx is None:
        dx = torch.empty_like(x)
    if want_gres and gres is None:
        gres = torch.empty_like(x)
    lib.pfr_bn_bwd_apply(_p(dout), _p(out_act), _p(x), _p(coef), _p(scale), _p(shift), mask_mode, _p(dx), _p(gres), did, rows,
                         C, _stream())
    return dx, gres, dgamma, dbeta


# ------------------------------------------------------------------------------------------------ pooling
def bn_relu_maxpool_fwd(x, scale, shift, relu=True, want_idx=True):
    N, H, W, C = x.shape
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((N, OH, OW, C), dtype=x.dtype, device=x.device)
    idx = torch.empty((N, OH, OW, C), dtype=torch.uint8, device=x.device) if want_idx else None
    lib.pfr_bn_relu_maxpool_fwd(_p(x), _p(scale), _p(shift), _p(y), _p(idx), dtype_id(x.dtype), N, H, W, C, int(relu), _stream())
    return y, idx


def maxpool_bwd(dy, idx, in_hw):
    N, OH, OW, C = dy.shape
    H, W = in_hw
    dz = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
    lib.pfr_maxpool_bwd(_p(dy), _p(idx), _p(dz), dtype_id(dy.dtype), N, H, W, C, _stream())
    return dz


def avgpool_fwd(x):
    N, H, W, C = x.shape
    y = torch.empty((N, C), dtype=x.dtype, device=x.device)
    lib.pfr_avgpool_fwd(_p(x), _p(y), dtype_id(x.dtype), N, H * W, C, _stream())
    return y


def avgpool_bwd(dy, hw):
    N, C = dy.shape
    H, W = hw
    dx = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
    lib.pfr_avgpool_bwd(_p(dy), _p(dx), dtype_id(dy.dtype), N, H * W, C, _stream())
    return dx


# ------------------------------------------------------------------------------------------------ head
def l2norm_fwd(x, out_dtype, want_t=False, ldt=None, xn=None, xnT=None, inv=None):
    rows, D = x.shape
    if xn is None:
        xn = torch.empty((rows, D), dtype=out_dtype, device=x.device)
    if want_t and xnT is None:
        xnT = torch.zeros((D, ldt or rows), dtype=out_dtype, device=x.device)
    if inv is None:
        inv = torch.empty(rows, dtype=torch.float32, device=x.device)
    lib.pfr_l2norm_fwd(_p(x), dtype_id(x.dtype), _p(xn), _p(xnT), dtype_id(out_dtype), _p(inv), rows, D,
                       0 if xnT is None else xnT.shape[1], 1e-12, _stream())
    return xn, xnT, inv


def l2norm_bwd(x, inv, dxn, out_dtype, out=None, accumulate=False):
    rows, D = x.shape
    assert dxn.dtype == torch.float32
    if out is None:
        out = torch.empty((rows, D), dtype=out_dtype, device=x.device)
    lib.pfr_l2norm_bwd(_p(x), dtype_id(x.dtype), _p(inv), _p(dxn), _p(out), dtype_id(out.dtype), rows, D, int(accumulate), _stream())
    return out


MARGIN_MODES = {"arc": 0, "arc_easy": 1, "cos": 2, "none": 3}


def margin_ce(cosv, label, C, mode, s, m, gamma=0.0, grad_scale=1.0, grad_scale_dev=None, want_logits=True, dcos_dtype=None, dcos=None,
              logits=None, loss_rows=None):
    B, ldc = cosv.shape
    assert cosv.dtype == torch.float32 and label.dtype == torch.int64
    if want_logits and logits is None:
        logits = torch.empty((B, C), dtype=torch.float32, device=cosv.device)
    if loss_rows is None:
        loss_rows = torch.empty(B, dtype=torch.float32, device=cosv.device)
    if dcos_dtype is not None and dcos is None:
        dcos = torch.zeros((B, ldc), dtype=dcos_dtype, device=cosv.device)
    lib.pfr_margin_ce(_p(cosv), _p(label), B, C, ldc, MARGIN_MODES[mode], float(s), float(m), float(gamma), float(grad_scale),
                      _p(grad_scale_dev), _p(logits), _p(loss_rows), _p(dcos), PFR_F32 if dcos is None else dtype_id(dcos.dtype), _stream())
    return logits, loss_rows, dcos


def mean(x, out=None):
    if out is None:
        out = torch.empty((), dtype=torch.float32, device=x.device)
    lib.pfr_mean(_p(x), _p(out), x.numel(), _stream())
    return out


# ------------------------------------------------------------------------------------------------ optimisers
def sgd_step(p, g, mom, shadow, lr, momentum, weight_decay, grad_scale=1.0, first_step=False):
    lib.pfr_sgd_step(_p(p), _p(g), _p(mom), _p(shadow), PFR_F32 if shadow is None else dtype_id(shadow.dtype), p.numel(),
                     float(lr), float(momentum), float(weight_decay), float(grad_scale), int(first_step), _stream())


def adamw_step(p, g, m, v, shadow, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    lib.pfr_adamw_step(_p(p), _p(g), _p(m), _p(v), _p(shadow), PFR_F32 if shadow is None else dtype_id(shadow.dtype), p.numel(),
                       float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale),
                       _stream())
