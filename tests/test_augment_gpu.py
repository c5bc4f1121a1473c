"""Device augmentation (csrc/pfr_augment.hip through the C-ABI) vs the Pillow-pinned oracle and the PIL-produced golden
fixture: bit-exact, per op and as the whole fe_dogs_config.py:17-26 pipeline."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import augment_ref as A
from test_augment_oracle import _images

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def test_augment_params_host_records_equal_oracle_matrices():
    """pfr_augment_params is host code (no device work): Pillow's rotate matrix in 16.16 fixed point"""
    from pets_face_recognition_amd._hip import lib
    rs = np.random.RandomState(0)
    angles = np.concatenate([rs.uniform(-5, 5, 500), [0.0, -0.0, 5.0, -5.0, 1e-6, -1e-6, 37.5, 359.0, -361.25]]).astype(np.float32)
    n = len(angles)
    flags = rs.randint(0, 5, (n, 4)).astype(np.int32)
    for (w, h) in ((224, 224), (48, 52), (301, 117)):
        rec = np.zeros((n, 12), np.int32)
        lib.pfr_augment_params(flags.ctypes.data, angles.ctypes.data, n, w, h, rec.ctypes.data)
        assert np.array_equal(rec[:, :4], flags)
        for i in range(n):
            fx = A.rotate_matrix_fixed(float(angles[i]), w, h)
            if fx is None:
                assert rec[i, 4] == 0
            else:
                assert rec[i, 4] == 1 and tuple(rec[i, 5:11]) == fx, (angles[i], rec[i], fx)


def _run(aug, x, flags, angles):
    y = aug.apply(torch.from_numpy(x).to(DEV), torch.from_numpy(flags), torch.from_numpy(angles))
    torch.cuda.synchronize()
    return y.cpu()


def _want(x, flags, angles, crop, out):
    res = []
    for i in range(x.shape[0]):
        img = x[i]
        if flags[i, 0]:
            img = A.smooth(img)
        if flags[i, 1]:
            img = A.autocontrast(img)
        img = img[flags[i, 2]:flags[i, 2] + crop[0], flags[i, 3]:flags[i, 3] + crop[1]]
        img = A.resize_bilinear(img, out[0], out[1])
        img = A.rotate_nearest(img, float(angles[i]))
        res.append(A.to_tensor(img))
    return torch.stack(res)


@pytest.mark.gpu
def test_pipeline_equals_pillow_golden():
    from pets_face_recognition_amd.data_loading import DeviceAugmentation
    z = np.load(os.path.join(GOLD, "augment.npz"))
    for tag in ("full", "small"):
        crop, size = int(z[f"{tag}_crop"]), int(z[f"{tag}_size"])
        aug = DeviceAugmentation((crop, crop), (size, size))
        y = _run(aug, z[f"{tag}_x"], z[f"{tag}_flags"], z[f"{tag}_angles"])
        want = torch.from_numpy(z[f"{tag}_out"].transpose(0, 3, 1, 2).copy()).float() / 255
        assert y.shape == want.shape and torch.equal(y, want), (tag, (y != want).float().mean())


@pytest.mark.gpu
def test_each_op_alone_equals_oracle():
    from pets_face_recognition_amd.data_loading import DeviceAugmentation
    for H, W in ((224, 224), (61, 47)):
        x = np.stack(_images(21 + H, 8, H, W))
        n = x.shape[0]
        zero = np.zeros((n, 4), np.int32)
        noang = np.zeros(n, np.float32)
        ident = DeviceAugmentation(None, None, 0, 0, 0)
        # ToTensor only (val_augmentation)
        assert torch.equal(_run(ident, x, zero, noang), _want(x, zero, noang, (H, W), (H, W)))
        # sharpness 0 (SMOOTH) alone; autocontrast alone (incl. a flat band: lo == hi → identity); both
        for s, c in ((1, 0), (0, 1), (1, 1)):
            f = zero.copy()
            f[:, 0], f[:, 1] = s, c
            assert torch.equal(_run(ident, x, f, noang), _want(x, f, noang, (H, W), (H, W))), (H, W, s, c)
        # crop + resize alone: up-scaling (the 220 → 224 of the configs), mixed, 2.3x down-scaling (5 taps)
        for (ch, cw), (oh, ow) in (((H - 4, W - 4), (H, W)), ((H - 4, W - 7), (H - 9, W + 6)), ((H, W), (H * 10 // 23, W * 10 // 23))):
            f = zero.copy()
            f[:, 2] = np.arange(n) % (H - ch + 1)
            f[:, 3] = (3 * np.arange(n)) % (W - cw + 1)
            aug = DeviceAugmentation((ch, cw), (oh, ow), 0, 0, 0)
            assert torch.equal(_run(aug, x, f, noang), _want(x, f, noang, (ch, cw), (oh, ow))), (H, W, ch, cw, oh, ow)
        # rotation alone
        ang = np.array([-5, 5, 0, 1e-3, -2.5, 4.999, 0.3, -0.01], np.float32)
        assert torch.equal(_run(ident, x, zero, ang), _want(x, zero, ang, (H, W), (H, W)))


@pytest.mark.gpu
def test_random_batches_equal_oracle_and_config_variants():
    """whole pipeline on random decisions: the 224/220/224 configs and the 256/252/224 variant (configs/cat_fe)"""
    from pets_face_recognition_amd.data_loading import DeviceAugmentation, train_augmentation
    for H, crop in ((224, 220), (256, 252)):
        x = np.stack(_images(7 + H, 24, H, H))
        aug = DeviceAugmentation((crop, crop), (224, 224), 0.4, 0.5, 5.0, generator=torch.Generator().manual_seed(H))
        flags, angles = aug.draw(x.shape[0], H, H)
        assert flags[:, 0].any() and flags[:, 1].any() and flags[:, 2].max() <= H - crop
        y = _run(aug, x, flags.numpy(), angles.numpy())
        assert torch.equal(y, _want(x, flags.numpy(), angles.numpy(), (crop, crop), (224, 224)))
    y = train_augmentation(torch.Generator().manual_seed(1))(torch.from_numpy(np.stack(_images(1, 4, 224, 224))).to(DEV))
    assert y.shape == (4, 3, 224, 224) and y.dtype == torch.float32 and 0 <= float(y.min()) and float(y.max()) <= 1
