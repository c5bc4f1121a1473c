"""HIP execution of the cosine-margin head and the (focal) cross-entropy — autograd plumbing around the C-ABI.

Replaces, for CUDA tensors, the torch ops of /root/reference/losses/large_margin.py:30-40,69-84 and
/root/reference/losses/losses.py:22-28 (see csrc/pfr_head.hip for the kernel-level mapping)."""
import torch

from .._hip import ops, PfrError
from ..models._fe_engine import default_compute_dtype


def _cpad(C, dtype):
    k = 8 if dtype == torch.bfloat16 else 4
    return (C + k - 1) // k * k


def _dxn_by_rows(B, Cp, T):
    """dx̂ as a row-split reduction over the classes (see _cosine_bwd) rather than a plain GEMM against ŵᵀ"""
    return B % (8 if T == torch.bfloat16 else 4) == 0 and Cp >= 2048


def _cosine_fwd(emb, weight, T):
    """→ cos [B, Cpad] f32 and the saved normalised operands"""
    B, D = emb.shape
    C = weight.shape[0]
    Cp = _cpad(C, T)
    xn, _, inv_x = ops.l2norm_fwd(emb, T)
    wn_full = torch.zeros((Cp, D), dtype=T, device=emb.device) if Cp != C else torch.empty((C, D), dtype=T, device=emb.device)
    # ŵᵀ is the operand of the plain-GEMM form of dx̂ only: _cosine_bwd's row-split form reads ŵ itself (and the transposed
    # write, 2-byte elements a class apart, is most of this kernel's time at 10 000 classes)
    wn, wnT, inv_w = ops.l2norm_fwd(weight, T, want_t=not _dxn_by_rows(B, Cp, T), ldt=Cp, xn=wn_full[:C])
    cos = torch.empty((B, 1, 1, Cp), dtype=torch.float32, device=emb.device)
    ops.conv2d_fwd(xn.view(B, 1, 1, D), wn.view(C, 1, 1, D), out=cos)
    return cos.view(B, Cp), (xn, wnT, inv_x, inv_w, wn_full)


def _cosine_bwd(dcos, emb, weight, saved, T):
    """dcos [B, Cpad] (compute dtype) → (demb f32 [B,D], dweight f32 [C,D])"""
    xn, wnT, inv_x, inv_w, wn_full = saved
    B, D = emb.shape
    C = weight.shape[0]
    Cp = dcos.shape[1]
    if _dxn_by_rows(B, Cp, T):
        # dxn[b][d] = Σ_c dcos[b][c]·wn[c][d]: the reduction runs over the CLASS dimension (10 000), the output is only B x D —
        # as a plain GEMM that is 8-16 tiles with a 10 000-long k-loop (0.2 ms on 8 CUs); as a "weight gradient" over the rows
        # of (wn, dcosᵀ) it is split over ~40 row ranges and takes ~20 µs
        from .._hip import lib, dtype_id
        dcosT = torch.empty((Cp, B), dtype=dcos.dtype, device=dcos.device)
        lib.pfr_transpose2d(dcos.data_ptr(), dcosT.data_ptr(), dtype_id(dcos.dtype), B, Cp, torch.cuda.current_stream().cuda_stream)
        dxn = ops.conv2d_wgrad(wn_full.view(Cp, 1, 1, D), dcosT.view(Cp, 1, 1, B), 1, 1, 1, 0)   # [B,1,1,D] f32
    else:
        dxn = torch.empty((B, 1, 1, D), dtype=torch.float32, device=emb.device)
        ops.conv2d_fwd(dcos.view(B, 1, 1, Cp), wnT.view(D, 1, 1, Cp), out=dxn)
    dwn = ops.conv2d_wgrad(xn.view(B, 1, 1, D), dcos.view(B, 1, 1, Cp), 1, 1, 1, 0)  # [Cp,1,1,D] f32
    demb = ops.l2norm_bwd(emb, inv_x, dxn.view(B, D), torch.float32)
    dw = ops.l2norm_bwd(weight, inv_w, dwn.view(Cp, D), torch.float32, out=torch.empty_like(weight))
    return demb, dw


class MarginCEFunction(torch.autograd.Function):
    """(emb, weight, label) → (loss, logits): normalise → cosine GEMM → margin → scale → softmax-CE, fused."""

    @staticmethod
    def forward(ctx, emb, weight, label, mode, s, m, gamma, T, want_logits):
        emb = emb.contiguous().float()
        w = weight.detach().contiguous()
        label = label.contiguous().long()
        C = w.shape[0]
        cos, saved = _cosine_fwd(emb, w, T)
        logits, loss_rows, _ = ops.margin_ce(cos, label, C, mode, s, m, gamma=gamma, want_logits=want_logits)
        loss = ops.mean(loss_rows)
        ctx.save_for_backward(emb, w, label, cos, *saved)
        ctx.cfg = (mode, s, m, gamma, T, C)
        if logits is None:
            logits = torch.empty(0, device=emb.device)
        ctx.mark_non_differentiable(logits)
        return loss, logits

    @staticmethod
    def backward(ctx, dloss, _dlogits):
        emb, w, label, cos, *saved = ctx.saved_tensors
        mode, s, m, gamma, T, C = ctx.cfg
        B = emb.shape[0]
        dloss = dloss.contiguous().float()
        _, _, dcos = ops.margin_ce(cos, label, C, mode, s, m, gamma=gamma, grad_scale=1.0 / B, grad_scale_dev=dloss,
                                   want_logits=False, dcos_dtype=T)
        demb, dw = _cosine_bwd(dcos, emb, w, saved, T)
        return demb, dw, None, None, None, None, None, None, None


class MarginFunction(torch.autograd.Function):
    """(emb, weight, label) → logits; standalone ArcMarginProduct / AddMarginProduct."""

    @staticmethod
    def forward(ctx, emb, weight, label, mode, s, m, T):
        emb = emb.contiguous().float()
        w = weight.detach().contiguous()
        label = label.contiguous().long()
        C = w.shape[0]
        cos, saved = _cosine_fwd(emb, w, T)
        logits, _, _ = ops.margin_ce(cos, label, C, mode, s, m, want_logits=True)
        ctx.save_for_backward(emb, w, label, cos, *saved)
        ctx.cfg = (mode, s, m, T, C)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        from .._hip import lib, dtype_id
        emb, w, label, cos, *saved = ctx.saved_tensors
        mode, s, m, T, C = ctx.cfg
        B, Cp = cos.shape
        dlogits = dlogits.contiguous().float()
        dcos = torch.zeros((B, Cp), dtype=T, device=emb.device)
        lib.pfr_margin_bwd(cos.data_ptr(), label.data_ptr(), B, C, Cp, ops.MARGIN_MODES[mode], float(s), float(m),
                           dlogits.data_ptr(), dcos.data_ptr(), dtype_id(T), torch.cuda.current_stream().cuda_stream)
        demb, dw = _cosine_bwd(dcos, emb, w, saved, T)
        return demb, dw, None, None, None, None, None


class FocalCEFunction(torch.autograd.Function):
    """(logits, target) → mean((1-p)^γ · CE): standalone FocalLoss / CrossEntropyLoss on CUDA logits."""

    @staticmethod
    def forward(ctx, logits, target, gamma):
        logits = logits.contiguous().float()
        target = target.contiguous().long()
        B, C = logits.shape
        _, rows, _ = ops.margin_ce(logits, target, C, "none", 1.0, 0.0, gamma=gamma, want_logits=False)
        ctx.save_for_backward(logits, target)
        ctx.gamma = gamma
        return ops.mean(rows)

    @staticmethod
    def backward(ctx, dloss):
        logits, target = ctx.saved_tensors
        B, C = logits.shape
        _, _, d = ops.margin_ce(logits, target, C, "none", 1.0, 0.0, gamma=ctx.gamma, grad_scale=1.0 / B,
                                grad_scale_dev=dloss.contiguous().float(), want_logits=False, dcos_dtype=torch.float32)
        return d, None, None


def resolve_dtype(dt):
    return dt or default_compute_dtype()
