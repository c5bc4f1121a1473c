"""Cosine-margin classification heads with the reference's constructor signatures and parameter names.

API mirrored: /root/reference/losses/large_margin.py — `AddMarginProduct(in_features, out_features, s=30.0, m=0.40)`
(CosFace, lines 10-40) and `ArcMarginProduct(in_features, out_features, s=30.0, m=0.50, easy_margin=False)`
(ArcFace, lines 44-84); `.weight` is `(out_features, in_features)`, Xavier-uniform.

CUDA inputs run on the gfx950 kernels (losses/_head_hip.py); CPU inputs run the same arithmetic with torch ops."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _MarginHead(nn.Module):
    _mode = None

    def __init__(self, in_features, out_features, s, m):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.s, self.m = s, m
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.xavier_uniform_(self.weight)
        self.compute_dtype = None  # HIP compute dtype (None → PFR_COMPUTE_DTYPE / bf16)

    def hip_mode(self):
        return self._mode

    def _target_logit(self, cosine):  # torch (CPU) formulation of the margin on every entry
        raise NotImplementedError

    def forward(self, input, label):
        if input.is_cuda:
            from ._head_hip import MarginFunction, resolve_dtype
            return MarginFunction.apply(input, self.weight, label, self.hip_mode(), self.s, self.m,
                                        resolve_dtype(self.compute_dtype))
        cosine = F.normalize(input) @ F.normalize(self.weight).t()
        target = self._target_logit(cosine)
        hot = F.one_hot(label.view(-1).long(), self.out_features).to(cosine.dtype)
        return self.s * (hot * target + (1.0 - hot) * cosine)


class AddMarginProduct(_MarginHead):
    """CosFace: s·(cos θ − m) on the target class."""
    _mode = "cos"

    def __init__(self, in_features, out_features, s=30.0, m=0.40, device=None, **_):
        super().__init__(in_features, out_features, s, m)
        self.device = device

    def _target_logit(self, cosine):
        return cosine - self.m


class ArcMarginProduct(_MarginHead):
    """ArcFace: s·cos(θ + m) on the target class, with the hard (default) or easy fallback outside [0, π−m]."""

    def __init__(self, in_features, out_features, s=30.0, m=0.50, easy_margin=False, **_):
        super().__init__(in_features, out_features, s, m)
        self.easy_margin = easy_margin
        self.cos_m, self.sin_m = math.cos(m), math.sin(m)
        self.th = math.cos(math.pi - m)
        self.mm = math.sin(math.pi - m) * m

    def hip_mode(self):
        return "arc_easy" if self.easy_margin else "arc"

    def _target_logit(self, cosine):
        sine = torch.sqrt(1.0 - cosine * cosine)
        shifted = cosine * self.cos_m - sine * self.sin_m
        if self.easy_margin:
            return torch.where(cosine > 0, shifted, cosine)
        return torch.where(cosine > self.th, shifted, cosine - self.mm)
