"""CPU restatement (numpy, integer / IEEE-double arithmetic) of the reference's TRAIN-time image augmentation
(/root/reference/configs/dog_fe/fe_dogs_config.py:17-26):

    ToPILImage → RandomAdjustSharpness(0, p=0.1) → RandomAutocontrast(p=0.3) → RandomCrop((220, 220))
               → Resize((224, 224)) → RandomRotation(5) → ToTensor

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke()); the product path is pets_face_recognition_amd/data_loading/
augment.py over csrc/pfr_augment.hip.

The algorithms live in third-party dependencies that are not vendored in /root/reference: torchvision (requirements.txt:3,
`torchvision>=0.12.0`, NOT installed in this image) and Pillow (requirements.txt:6, `Pillow==9.0.1`; 12.2.0 is installed
here).  On PIL images torchvision's transforms are thin calls into Pillow (torchvision/transforms/functional_pil.py):
    adjust_sharpness(img, f)   = ImageEnhance.Sharpness(img).enhance(f)          (f = 0 → img.filter(ImageFilter.SMOOTH))
    autocontrast(img)          = ImageOps.autocontrast(img)
    crop(img, i, j, h, w)      = img.crop((j, i, j + w, i + h))
    resize(img, (h, w))        = img.resize((w, h), BILINEAR)
    rotate(img, angle)         = img.rotate(angle, NEAREST, expand=False, center=None, fillcolor=(0, 0, 0))
    to_tensor(img)             = uint8 HWC → float32 CHW / 255
so the restatement follows Pillow's published C / Python (ImageFilter.SMOOTH + Filter.c 3x3; ImageOps.autocontrast;
Resample.c precompute_coeffs / normalize_coeffs_8bpc / 8bpc passes; Image.rotate + Geometry.c affine_fixed) and is PINNED
against the installed Pillow itself: tests/test_augment_oracle.py runs every function below against the PIL call above on
random images, and tests/golden/augment.npz holds PIL-produced outputs (oracle/make_golden.py:gen_augment).

The random DECISIONS (apply flags, crop corner, angle) are inputs here: in the reference they are drawn per sample inside
dataloader worker processes, so only "same decisions → same pixels" is a meaningful parity statement.
`draw_params` restates the order and distributions torchvision draws them in.
"""
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2          # Resample.c


# ------------------------------------------------------------------------------------------------ sharpness factor 0
def smooth(img):
    """ImageFilter.SMOOTH: 3x3 kernel (1,1,1,1,5,1,1,1,1)/13, offset 0; Filter.c leaves the 1-pixel frame untouched and
    rounds with +0.5 then truncation.  img uint8 [H, W, C].  (The weighted sum is an integer S, and S/13 + 0.5 is never
    within 0.038 of an integer, so floor((2S + 13) / 26) equals Pillow's float32 evaluation exactly.)"""
    a = img.astype(np.int64)
    out = img.copy()
    H, W = img.shape[:2]
    if H < 3 or W < 3:
        return out
    S = np.zeros((H - 2, W - 2) + img.shape[2:], np.int64)
    for dy in range(3):
        for dx in range(3):
            S += a[dy:dy + H - 2, dx:dx + W - 2] * (5 if dy == 1 and dx == 1 else 1)
    out[1:-1, 1:-1] = np.clip((2 * S + 13) // 26, 0, 255).astype(np.uint8)
    return out


# ------------------------------------------------------------------------------------------------------ autocontrast
def autocontrast_lut(lo, hi):
    """ImageOps.autocontrast (cutoff 0): per band, lo / hi = darkest / brightest value present; identity if hi <= lo,
    else lut[ix] = clip(int(ix * scale + offset)) with scale = 255.0 / (hi - lo), offset = -lo * scale (Python doubles,
    int() truncates toward zero)."""
    if hi <= lo:
        return np.arange(256, dtype=np.uint8)
    scale = 255.0 / (hi - lo)
    offset = -lo * scale
    lut = np.empty(256, np.uint8)
    for ix in range(256):
        v = int(ix * scale + offset)
        lut[ix] = 0 if v < 0 else (255 if v > 255 else v)
    return lut


def autocontrast(img):
    out = np.empty_like(img)
    for c in range(img.shape[2]):
        band = img[..., c]
        out[..., c] = autocontrast_lut(int(band.min()), int(band.max()))[band]
    return out


# ------------------------------------------------------------------------------------------------------------ resize
def resize_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter over the box
    (0, in_size).  → (bounds int32 [out, 2] = (first tap, tap count), kk int32 [out, ksize])"""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            w.append(1.0 - t if t < 1.0 else 0.0)
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis0(img, out_size):
    bounds, kk = resize_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    a = img.astype(np.int64)
    for yy in range(out_size):
        y0, n = bounds[yy]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for t in range(n):
            acc += a[y0 + t] * int(kk[yy, t])
        out[yy] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bilinear(img, out_h, out_w):
    """img.resize((out_w, out_h), BILINEAR): horizontal pass to an 8-bit image, then vertical pass (Resample.c
    ImagingResampleInner); a pass whose size is unchanged is skipped."""
    if img.shape[1] != out_w:
        img = _resample_axis0(img.transpose(1, 0, 2), out_w).transpose(1, 0, 2)
    if img.shape[0] != out_h:
        img = _resample_axis0(img, out_h)
    return np.ascontiguousarray(img)


# ------------------------------------------------------------------------------------------------------------ rotate
def rotate_matrix_fixed(angle, w, h):
    """Image.rotate (expand=False, center=None) → Image.transform(AFFINE) → Geometry.c affine_fixed: the 16.16 fixed-point
    coefficients (a0, a1, a2, a3, a4, a5); source x of output (x, y) = (a2 + a0*x + a1*y) >> 16, source y likewise with
    (a5, a3, a4).  → None when Pillow takes a shortcut that copies the image (angle % 360 == 0)."""
    angle = angle % 360.0
    if angle == 0:
        return None
    if angle in (90.0, 180.0, 270.0) and (angle == 180.0 or w == h):
        raise NotImplementedError("right-angle rotations use Image.transpose in Pillow; outside RandomRotation(5)")
    cx, cy = w / 2.0, h / 2.0
    ang = -math.radians(angle)
    m = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0, round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy

    def fix(v):
        v = v * 65536.0 + 0.5
        return int(v) if v >= 0.0 else int(math.floor(v))
    return (fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5))


def rotate_nearest(img, angle):
    H, W = img.shape[:2]
    fx = rotate_matrix_fixed(angle, W, H)
    if fx is None:
        return img.copy()
    a0, a1, a2, a3, a4, a5 = fx
    y, x = np.mgrid[0:H, 0:W].astype(np.int64)
    xin = (a2 + a0 * x + a1 * y) >> 16
    yin = (a5 + a3 * x + a4 * y) >> 16
    ok = (xin >= 0) & (xin < W) & (yin >= 0) & (yin < H)
    out = np.zeros_like(img)
    out[ok] = img[yin[ok], xin[ok]]
    return out


# ---------------------------------------------------------------------------------------------------------- pipeline
def to_tensor(img):
    """torchvision ToTensor: uint8 HWC → float32 CHW, divided by 255 (float32 division)"""
    return torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).float().div(255)


def draw_params(n, H, W, crop=220, p_sharp=0.1, p_contrast=0.3, degrees=5.0, generator=None):
    """The random decisions of one pass over n images, drawn in torchvision's order per image (RandomAdjustSharpness:
    torch.rand(1) < p; RandomAutocontrast: torch.rand(1) < p; RandomCrop.get_params: randint(0, H-crop+1) then
    randint(0, W-crop+1); RandomRotation.get_params: empty(1).uniform_(-d, d)) — int32 [n, 4] (sharp, contrast, top,
    left) and float32 [n] angles."""
    g = generator
    flags = np.zeros((n, 4), np.int32)
    angles = np.zeros(n, np.float32)
    for i in range(n):
        flags[i, 0] = int(torch.rand(1, generator=g).item() < p_sharp)
        flags[i, 1] = int(torch.rand(1, generator=g).item() < p_contrast)
        flags[i, 2] = int(torch.randint(0, H - crop + 1, (1,), generator=g).item())
        flags[i, 3] = int(torch.randint(0, W - crop + 1, (1,), generator=g).item())
        angles[i] = torch.empty(1).uniform_(-degrees, degrees, generator=g).item()
    return flags, angles


def train_augmentation(img, sharp, contrast, top, left, angle, crop=220, out=224):
    """one image through fe_dogs_config.py:17-26 with the given decisions → (uint8 [out, out, C], float32 [C, out, out])"""
    if sharp:
        img = smooth(img)
    if contrast:
        img = autocontrast(img)
    img = img[top:top + crop, left:left + crop]
    img = resize_bilinear(img, out, out)
    img = rotate_nearest(img, float(angle))
    return img, to_tensor(img)
