"""Device-side image augmentation: the reference's `train_augmentation` / `val_augmentation` Compose pipelines
(/root/reference/configs/dog_fe/fe_dogs_config.py:17-32) applied to a whole uint8 batch on the GPU.

    train:  ToPILImage → RandomAdjustSharpness(0, 0.1) → RandomAutocontrast(0.3) → RandomCrop((220, 220)) →
            Resize((224, 224)) → RandomRotation(5) → ToTensor
    val:    ToPILImage → [Resize((224, 224))] → ToTensor

The reference runs these per sample on PIL images in dataloader worker processes; here the dataset hands over raw uint8
HWC frames (what `RecDataset.__getitem__` holds before the transform, data_loading/dataset.py:100-121), the batch is
uploaded once and csrc/pfr_augment.hip produces the float32 NCHW batch `batch['x']` the trainer consumes — same pixels,
bit for bit, as the PIL pipeline given the same random decisions (tests/test_augment_gpu.py).  The decisions themselves
are drawn here on the host with the distributions torchvision uses (Bernoulli(p) flags, uniform integer crop corner,
uniform angle); the reference draws them inside worker processes with per-worker seeds, so its stream is not reproducible
and is not part of the contract.  There is no CPU implementation: without the HIP library this module raises.
"""
import ctypes

import numpy as np
import torch

from .._hip import lib, PfrError

_REC = 12


def _stream():
    return torch.cuda.current_stream().cuda_stream


class DeviceAugmentation:
    """crop=None → no random crop (validation); size=None → no resize; p_* = 0 and degrees = 0 switch the others off."""

    def __init__(self, crop=(220, 220), size=(224, 224), p_sharpness=0.1, p_autocontrast=0.3, degrees=5.0, generator=None):
        self.crop = tuple(crop) if crop is not None else None
        self.size = tuple(size) if size is not None else None
        self.p_sharpness, self.p_autocontrast, self.degrees = float(p_sharpness), float(p_autocontrast), float(degrees)
        self.generator = generator
        self._ws = None

    def draw(self, n, H, W):
        """→ (flags int32 [n, 4] = (sharpness, autocontrast, top, left), angles float32 [n]) — host tensors"""
        g = self.generator
        ch, cw = self.crop if self.crop is not None else (H, W)
        if ch > H or cw > W:
            raise PfrError(f"DeviceAugmentation: crop {ch}x{cw} larger than the {H}x{W} input")
        flags = torch.zeros((n, 4), dtype=torch.int32)
        u = torch.rand((n, 2), generator=g)
        flags[:, 0] = (u[:, 0] < self.p_sharpness).int()
        flags[:, 1] = (u[:, 1] < self.p_autocontrast).int()
        flags[:, 2] = torch.randint(0, H - ch + 1, (n,), generator=g).int()
        flags[:, 3] = torch.randint(0, W - cw + 1, (n,), generator=g).int()
        angles = torch.empty(n).uniform_(-self.degrees, self.degrees, generator=g) if self.degrees > 0 else torch.zeros(n)
        return flags, angles

    def apply(self, x, flags, angles):
        """x uint8 [N, H, W, 3] on the GPU; flags / angles as `draw` returns them → float32 [N, 3, out_h, out_w]"""
        if not x.is_cuda or x.dtype != torch.uint8 or x.dim() != 4 or x.shape[3] != 3:
            raise PfrError("DeviceAugmentation: expects a uint8 [N, H, W, 3] CUDA batch")
        x = x.contiguous()
        N, H, W, _ = x.shape
        ch, cw = self.crop if self.crop is not None else (H, W)
        oh, ow = self.size if self.size is not None else (ch, cw)
        flags = np.ascontiguousarray(torch.as_tensor(flags).numpy(), dtype=np.int32).reshape(N, 4)
        angles = np.ascontiguousarray(torch.as_tensor(angles).numpy(), dtype=np.float32).reshape(N)
        if (flags[:, 2] < 0).any() or (flags[:, 2] + ch > H).any() or (flags[:, 3] < 0).any() or (flags[:, 3] + cw > W).any():
            raise PfrError("DeviceAugmentation: crop window outside the image")
        rec = torch.empty((N, _REC), dtype=torch.int32).pin_memory()
        lib.pfr_augment_params(flags.ctypes.data, angles.ctypes.data, N, ow, oh, rec.data_ptr())
        rec_d = rec.to(x.device, non_blocking=True)
        need = lib.pfr_augment_ws_bytes(N, H, W)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        y = torch.empty((N, 3, oh, ow), dtype=torch.float32, device=x.device)
        lib.pfr_augment_train(x.data_ptr(), N, H, W, ch, cw, oh, ow, rec_d.data_ptr(), y.data_ptr(), self._ws.data_ptr(), _stream())
        return y

    def __call__(self, x):
        flags, angles = self.draw(x.shape[0], x.shape[1], x.shape[2])
        return self.apply(x, flags, angles)


def train_augmentation(generator=None):
    """fe_dogs_config.py:17-26"""
    return DeviceAugmentation((220, 220), (224, 224), 0.1, 0.3, 5.0, generator)


def val_augmentation(size=None):
    """fe_dogs_config.py:28-32 (ToTensor only; six of the ten FE configs add Resize((224, 224)))"""
    return DeviceAugmentation(None, size, 0.0, 0.0, 0.0)
