"""Pins oracle/augment_ref.py against the installed Pillow (the library torchvision's PIL-image transforms call into) and
against tests/golden/augment.npz (PIL-produced, oracle/make_golden.py:gen_augment).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import augment_ref as A

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _images(seed, n, H, W):
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        yy, xx = np.mgrid[0:H, 0:W]
        base = np.stack([127 + 90 * np.sin(xx / (7.0 + i) + c) * np.cos(yy / (5.0 + c)) for c in range(3)], -1)
        noise = rs.randn(H, W, 3) * (3 + 20 * (i % 3))
        lo, hi = [(0, 255), (30, 200), (5, 250), (90, 91)][i % 4]
        out.append(np.clip(base + noise, lo, hi).astype(np.uint8))
    return out


def test_each_op_equals_pillow():
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageEnhance, ImageOps
    for H, W in ((224, 224), (61, 47), (230, 300)):
        for i, img in enumerate(_images(3 + H, 6, H, W)):
            pil = Image.fromarray(img)
            assert np.array_equal(A.smooth(img), np.asarray(ImageEnhance.Sharpness(pil).enhance(0)))
            assert np.array_equal(A.autocontrast(img), np.asarray(ImageOps.autocontrast(pil)))
            for oh, ow in ((H + 4, W + 4), (H - 9, W + 13), (224, 224)):
                assert np.array_equal(A.resize_bilinear(img, oh, ow), np.asarray(pil.resize((ow, oh), Image.BILINEAR)))
            for ang in (-5.0, -1.2345678, 0.0, 0.3, 4.999, float(np.float32(2.7182817))):
                assert np.array_equal(A.rotate_nearest(img, ang), np.asarray(pil.rotate(ang, Image.NEAREST, fillcolor=(0, 0, 0))))


def test_pipeline_equals_golden():
    z = np.load(os.path.join(GOLD, "augment.npz"))
    for tag in ("full", "small"):
        x, flags, angles, want = z[f"{tag}_x"], z[f"{tag}_flags"], z[f"{tag}_angles"], z[f"{tag}_out"]
        crop, out = int(z[f"{tag}_crop"]), int(z[f"{tag}_size"])
        assert flags[:, 0].any() and flags[:, 1].any() and (flags[:, :2].sum(1) == 2).any() and (flags[:, :2].sum(1) == 0).any()
        for i in range(x.shape[0]):
            u8, t = A.train_augmentation(x[i], *flags[i], angles[i], crop=crop, out=out)
            assert np.array_equal(u8, want[i])
            assert torch.equal(t, torch.from_numpy(want[i].transpose(2, 0, 1).copy()).float() / 255)


def test_draw_params_order_and_ranges():
    g = torch.Generator().manual_seed(5)
    flags, ang = A.draw_params(2000, 224, 224, generator=g)
    assert 0.07 < flags[:, 0].mean() < 0.13 and 0.26 < flags[:, 1].mean() < 0.34
    assert flags[:, 2:].min() == 0 and flags[:, 2:].max() == 4 and np.abs(ang).max() <= 5.0
    g2 = torch.Generator().manual_seed(5)
    assert (torch.rand(1, generator=g2).item() < 0.1) == bool(flags[0, 0])
