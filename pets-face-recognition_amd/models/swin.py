"""Swin Transformer with the module tree / state-dict names of /root/reference/models/swin.py (berniwal variant:
Unfold+Linear patch merging, pre-norm W-MSA / SW-MSA with cyclic shift and additive −inf masks on the last window
row / column, one shared (2w−1)² relative-position table, exact-GELU MLP, mean-pool → LayerNorm → Linear head).

Restated with plain reshape/permute (no einops).  CPU tensors run these torch layers; CUDA (HIP) tensors run the
gfx950 kernels through models/_swin_engine.SwinEngine (LayerNorm, GELU, fused shifted-window attention in
csrc/pfr_swin.hip; every Linear incl. patch merging on the MFMA GEMM kernels) — SURVEY.md §8 K14-K17, BASELINE config 4.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _shift_mask(w, d, vertical):
    """additive mask (w², w²): −inf between the two halves a cyclic shift by d glues together"""
    idx = torch.arange(w * w)
    coord = (idx // w) if vertical else (idx % w)
    part = coord >= (w - d)
    m = torch.zeros(w * w, w * w)
    m[part[:, None] != part[None, :]] = float("-inf")
    return m


def _relative_index(w):
    ys, xs = torch.meshgrid(torch.arange(w), torch.arange(w), indexing="ij")
    pos = torch.stack([ys.reshape(-1), xs.reshape(-1)], dim=1)
    return pos[None, :, :] - pos[:, None, :] + (w - 1)


class _Named(nn.Module):
    """holder giving a sub-module the attribute name `fn` (the reference nests Residual(PreNorm(...)))"""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class Residual(_Named):
    def forward(self, x, **kw):
        return x + self.fn(x, **kw)


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x, **kw):
        return self.fn(self.norm(x), **kw)


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, dim))

    def forward(self, x):
        return self.net(x)


class WindowAttention(nn.Module):
    def __init__(self, dim, heads, head_dim, shifted, window_size, relative_pos_embedding):
        super().__init__()
        inner = head_dim * heads
        self.heads, self.scale, self.window_size = heads, head_dim ** -0.5, window_size
        self.relative_pos_embedding, self.shifted = relative_pos_embedding, shifted
        if shifted:
            d = window_size // 2
            self.displacement = d
            self.upper_lower_mask = nn.Parameter(_shift_mask(window_size, d, True), requires_grad=False)
            self.left_right_mask = nn.Parameter(_shift_mask(window_size, d, False), requires_grad=False)
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        if relative_pos_embedding:
            self.relative_indices = _relative_index(window_size)
            self.pos_embedding = nn.Parameter(torch.randn(2 * window_size - 1, 2 * window_size - 1))
        else:
            self.pos_embedding = nn.Parameter(torch.randn(window_size ** 2, window_size ** 2))
        self.to_out = nn.Linear(inner, dim)

    def _windows(self, t, b, gh, gw):
        w, h = self.window_size, self.heads
        t = t.view(b, gh, w, gw, w, h, -1).permute(0, 5, 1, 3, 2, 4, 6)
        return t.reshape(b, h, gh * gw, w * w, -1)

    def forward(self, x):
        w, h = self.window_size, self.heads
        if self.shifted:
            x = torch.roll(x, (-self.displacement, -self.displacement), (1, 2))
        b, nh, nw, _ = x.shape
        gh, gw = nh // w, nw // w
        q, k, v = (self._windows(t, b, gh, gw) for t in self.to_qkv(x).chunk(3, dim=-1))
        dots = torch.matmul(q, k.transpose(-1, -2)) * self.scale
        if self.relative_pos_embedding:
            ri = self.relative_indices
            dots = dots + self.pos_embedding[ri[:, :, 0], ri[:, :, 1]]
        else:
            dots = dots + self.pos_embedding
        if self.shifted:
            dots[:, :, -gw:] += self.upper_lower_mask
            dots[:, :, gw - 1::gw] += self.left_right_mask
        out = torch.matmul(dots.softmax(dim=-1), v)
        out = out.view(b, h, gh, gw, w, w, -1).permute(0, 2, 4, 3, 5, 1, 6).reshape(b, nh, nw, -1)
        out = self.to_out(out)
        if self.shifted:
            out = torch.roll(out, (self.displacement, self.displacement), (1, 2))
        return out


class SwinBlock(nn.Module):
    def __init__(self, dim, heads, head_dim, mlp_dim, shifted, window_size, relative_pos_embedding):
        super().__init__()
        self.attention_block = Residual(PreNorm(dim, WindowAttention(dim, heads, head_dim, shifted, window_size,
                                                                     relative_pos_embedding)))
        self.mlp_block = Residual(PreNorm(dim, FeedForward(dim, mlp_dim)))

    def forward(self, x):
        return self.mlp_block(self.attention_block(x))


class PatchMerging(nn.Module):
    def __init__(self, in_channels, out_channels, downscaling_factor):
        super().__init__()
        self.downscaling_factor = downscaling_factor
        self.patch_merge = nn.Unfold(kernel_size=downscaling_factor, stride=downscaling_factor, padding=0)
        self.linear = nn.Linear(in_channels * downscaling_factor ** 2, out_channels)

    def forward(self, x):
        b, c, hh, ww = x.shape
        f = self.downscaling_factor
        # space-to-depth with Unfold's (c, kh, kw) feature order
        x = x.view(b, c, hh // f, f, ww // f, f).permute(0, 2, 4, 1, 3, 5).reshape(b, hh // f, ww // f, c * f * f)
        return self.linear(x)


class StageModule(nn.Module):
    def __init__(self, in_channels, hidden_dimension, layers, downscaling_factor, num_heads, head_dim, window_size,
                 relative_pos_embedding):
        super().__init__()
        assert layers % 2 == 0, 'Stage layers need to be divisible by 2 for regular and shifted block.'
        self.patch_partition = PatchMerging(in_channels, hidden_dimension, downscaling_factor)
        self.layers = nn.ModuleList([
            nn.ModuleList([SwinBlock(hidden_dimension, num_heads, head_dim, hidden_dimension * 4, sh, window_size,
                                     relative_pos_embedding) for sh in (False, True)])
            for _ in range(layers // 2)])

    def forward(self, x):
        x = self.patch_partition(x)
        for regular, shifted in self.layers:
            x = shifted(regular(x))
        return x.permute(0, 3, 1, 2)


class SwinTransformer(nn.Module):
    def __init__(self, *, hidden_dim, layers, heads, channels=3, num_classes=1000, head_dim=32, window_size=7,
                 downscaling_factors=(4, 2, 2, 2), relative_pos_embedding=True):
        super().__init__()
        dims = [channels, hidden_dim, hidden_dim * 2, hidden_dim * 4, hidden_dim * 8]
        for i in range(4):
            setattr(self, f"stage{i + 1}", StageModule(dims[i], dims[i + 1], layers[i], downscaling_factors[i], heads[i],
                                                       head_dim, window_size, relative_pos_embedding))
        self.mlp_head = nn.Sequential(nn.LayerNorm(dims[4]), nn.Linear(dims[4], num_classes))
        self.compute_dtype = None   # HIP compute dtype: torch.bfloat16 / torch.float32 (None → PFR_COMPUTE_DTYPE / bf16)
        self._engine = None

    def _forward_torch(self, img):
        x = self.stage4(self.stage3(self.stage2(self.stage1(img))))
        return self.mlp_head(x.mean(dim=[2, 3]))

    def hip_engine(self, device=None):
        from ._swin_engine import SwinEngine
        if self._engine is None or not self._engine.matches(self):
            self._engine = SwinEngine(self, device or next(self.parameters()).device, self.compute_dtype)
        return self._engine

    def forward(self, img):
        if img.is_cuda:
            from ._swin_engine import swin_forward
            return swin_forward(self, img)
        return self._forward_torch(img)

    def _apply(self, fn, *a, **kw):
        self._engine = None
        return super()._apply(fn, *a, **kw)


def _build(hidden_dim, layers, heads, kwargs):
    dt = kwargs.pop("compute_dtype", None)
    m = SwinTransformer(hidden_dim=hidden_dim, layers=layers, heads=heads, **kwargs)
    m.compute_dtype = dt
    return m


def swin_t(hidden_dim=96, layers=(2, 2, 6, 2), heads=(3, 6, 12, 24), **kwargs):
    return _build(hidden_dim, layers, heads, kwargs)


def swin_s(hidden_dim=96, layers=(2, 2, 18, 2), heads=(3, 6, 12, 24), **kwargs):
    return _build(hidden_dim, layers, heads, kwargs)


def swin_b(hidden_dim=128, layers=(2, 2, 18, 2), heads=(4, 8, 16, 32), **kwargs):
    return _build(hidden_dim, layers, heads, kwargs)


def swin_l(hidden_dim=192, layers=(2, 2, 18, 2), heads=(6, 12, 24, 48), **kwargs):
    return _build(hidden_dim, layers, heads, kwargs)
