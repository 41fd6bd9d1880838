"""Host-side helpers of the SinDDM hot path (mirror of reference SinDDM/functions.py:72-192).

Everything here is one-off host work (schedule maths in float64 numpy, pyramid construction with
PIL) or tiny glue; the per-step work lives in the HIP library.
"""
from __future__ import annotations

from inspect import isfunction
from pathlib import Path
from typing import List, Sequence, Tuple

import numpy as np
import torch
from PIL import Image


def exists(x) -> bool:                                   # functions.py:72
    return x is not None


def default(val, d):                                     # functions.py:76
    if exists(val):
        return val
    return d() if isfunction(d) else d


def cycle(dl):                                           # functions.py:82
    while True:
        for data in dl:
            yield data


def num_to_groups(num: int, divisor: int) -> List[int]:  # functions.py:88
    groups, remainder = divmod(num, divisor)
    arr = [divisor] * groups
    if remainder > 0:
        arr.append(remainder)
    return arr


def loss_backwards(fp16, loss, optimizer, **kwargs):     # functions.py:97 (apex AMP is never enabled)
    if fp16:
        raise NotImplementedError("apex mixed precision is not part of the MI355X build (fp16=False in main.py:122)")
    loss.backward(**kwargs)


def extract(a: torch.Tensor, t: torch.Tensor, x_shape) -> torch.Tensor:
    """a[t] reshaped (B,1,1,1) -- functions.py:105-108.  Kept for API parity; the HIP kernels do
    this gather themselves (table pointer + t)."""
    b = t.shape[0]
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


def noise_like(shape, device, repeat: bool = False) -> torch.Tensor:   # functions.py:111-114
    if repeat:
        return torch.randn((1, *shape[1:]), device=device).repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)


def cosine_beta_schedule(timesteps: int, s: float = 0.008) -> np.ndarray:
    """Cosine schedule in float64 (functions.py:117-127)."""
    steps = timesteps + 1
    grid = np.linspace(0, steps, steps)
    abar = np.cos(((grid / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    abar = abar / abar[0]
    betas = 1 - (abar[1:] / abar[:-1])
    return np.clip(betas, a_min=0, a_max=0.999)


def pyramid_geometry(image_size: Tuple[int, int], scale_factor: float = 1.411, auto_scale=None):
    """Integer / float64 bookkeeping of create_img_scales (functions.py:148-174): returns
    (sizes[(W,H)], scale_factor, n_scales, image_size_used).  No image data involved."""
    image_size = (int(image_size[0]), int(image_size[1]))
    if auto_scale is not None:
        scaler = np.sqrt((image_size[0] * image_size[1]) / auto_scale)
        if scaler > 1:
            image_size = (int(image_size[0] / scaler), int(image_size[1] / scaler))
    area_scale_0 = 3110
    s_dim, l_dim = min(image_size), max(image_size)
    scale_0_dim = int(round(np.sqrt(area_scale_0 * s_dim / l_dim)))
    scale_0_dim = min(max(scale_0_dim, 42), 55)
    n_scales = int(round((np.log(s_dim / scale_0_dim)) / (np.log(scale_factor))) + 1)
    scale_factor = np.exp((np.log(s_dim / scale_0_dim)) / (n_scales - 1))
    sizes = []
    for i in range(n_scales):
        f = np.power(scale_factor, n_scales - i - 1)
        sizes.append((int(round(image_size[0] / f)), int(round(image_size[1] / f))))
    return sizes, scale_factor, n_scales, image_size


def create_img_scales(foldername, filename, scale_factor=1.411, image_size=None, create=False, auto_scale=None):
    """Build the image pyramid of one training image (functions.py:130-192).

    LANCZOS down-scales into <folder>/scale_i/, BILINEAR re-upsamples of scale i-1 into
    <folder>/scale_i_recon/, and the wrapped-uint8 Frobenius 'rescale loss' per scale.
    Returns (sizes[(W,H)], rescale_losses, scale_factor, n_scales) exactly like the reference."""
    orig_image = Image.open(foldername + filename)
    filename = filename.rsplit(".", 1)[0] + ".png"
    if image_size is None:
        image_size = orig_image.size
    sizes, scale_factor, n_scales, _ = pyramid_geometry(image_size, scale_factor, auto_scale)

    pyramid = []
    for i, size in enumerate(sizes):
        img = orig_image.resize(size, Image.LANCZOS)
        if create:
            out_dir = Path(foldername + f"scale_{i}/")
            out_dir.mkdir(parents=True, exist_ok=True)
            img.save(str(out_dir / filename))
        pyramid.append(img)

    rescale_losses = []
    for i in range(n_scales - 1):
        recon = pyramid[i].resize(sizes[i + 1], Image.BILINEAR)
        # uint8 subtraction wraps around, as in the reference (np.subtract on PIL images)
        diff = np.subtract(pyramid[i + 1], recon)
        rescale_losses.append(np.linalg.norm(diff) / np.asarray(recon).size)
        if create:
            out_dir = Path(foldername + f"scale_{i + 1}_recon/")
            out_dir.mkdir(parents=True, exist_ok=True)
            recon.save(str(out_dir / filename))
    return sizes, rescale_losses, scale_factor, n_scales


# ---- helpers of the application drivers (reference SinDDM/functions.py:21-48) ----------------------------
def extract_patch(image: torch.Tensor, bb) -> torch.Tensor:              # functions.py:45-48
    y_bb, x_bb, h_bb, w_bb = bb
    return image[:, :, y_bb:y_bb + h_bb, x_bb:x_bb + w_bb]


def stat_from_bbs(image: torch.Tensor, bb):                              # functions.py:38-42
    y_bb, x_bb, h_bb, w_bb = bb
    reg = image[:, :, y_bb:y_bb + h_bb, x_bb:x_bb + w_bb]
    return [torch.mean(reg, dim=(2, 3), keepdim=True), torch.std(reg, dim=(2, 3), keepdim=True)]


def _disk(radius: int) -> np.ndarray:
    """skimage.morphology.disk: (2r+1)^2 footprint of the pixels within Euclidean distance r."""
    yy, xx = np.mgrid[-radius:radius + 1, -radius:radius + 1]
    return (xx * xx + yy * yy) <= radius * radius


def dilate_mask(mask: torch.Tensor, mode: str) -> np.ndarray:
    """functions.py:21-33 -- binary dilation by a disk (radius 7 harmonization / 20 editing), Gaussian blur
    (sigma 5), min-max normalisation; returns (1,1,H,W).  The reference calls scikit-image (pinned 0.19.3; absent from this
    image's Python): restated on scipy.ndimage with scikit-image's defaults (binary_dilation: border value False;
    filters.gaussian: mode='nearest', truncate=4.0, float64) and PINNED by fixture G20 = the reference's lines run on
    scikit-image 0.18.3 itself (tests/golden/make_golden_skimage.py; agreement < 1e-12, tests/test_host.py)."""
    from scipy import ndimage as ndi
    if mode == "harmonization":
        element = _disk(7)
    elif mode == "editing":
        element = _disk(20)
    else:
        raise ValueError(mode)
    m = np.asarray(mask.permute(1, 2, 0)[:, :, 0]) != 0
    m = ndi.binary_dilation(m, structure=element)
    m = ndi.gaussian_filter(m.astype(np.float64), sigma=5, mode="nearest", truncate=4.0)
    m = m[None, None, :, :]
    return (m - m.min()) / (m.max() - m.min())


def match_histograms(image: np.ndarray, reference: np.ndarray, channel_axis: int = 2) -> np.ndarray:
    """skimage.exposure.match_histograms (0.19.3) for uint8 HxWxC images, as used by image2image
    (trainer.py:312-314): per channel, map every source level through the reference's inverse CDF
    (np.interp of the cumulative histograms) and store into the input dtype.  Pinned bit for bit by fixture G20 (scikit-image
    0.18.3's own output, tests/golden/make_golden_skimage.py)."""
    if image.ndim != reference.ndim or channel_axis != image.ndim - 1:
        raise ValueError("expects HxWxC arrays with the channel axis last")
    if image.shape[-1] != reference.shape[-1]:
        raise ValueError("Number of channels in the input image and reference image must match!")
    out = np.empty(image.shape, dtype=image.dtype)
    for ch in range(image.shape[-1]):
        src, tmpl = image[..., ch], reference[..., ch]
        lookup = src.reshape(-1)
        src_counts = np.bincount(lookup)
        tmpl_counts = np.bincount(tmpl.reshape(-1))
        tmpl_values = np.nonzero(tmpl_counts)[0]
        tmpl_counts = tmpl_counts[tmpl_values]
        src_q = np.cumsum(src_counts) / src.size
        tmpl_q = np.cumsum(tmpl_counts) / tmpl.size
        interp = np.interp(src_q, tmpl_q, tmpl_values)
        out[..., ch] = interp[lookup].reshape(src.shape)
    return out


def thresholded_grad(grad, quantile=0.8):
    """Soft-thresholded guidance gradient + the mask of the positions that survive (reference SinDDM/functions.py:52-67):
    per sample the pixel-wise gradient energy ||grad||_2 over channels is reduced by its `quantile` (nearest) and clamped
    at 0; the direction of the gradient is kept.  Returns (sparse_grad (B,C,H,W), mask (B,1,H,W) bool)."""
    energy = torch.norm(grad, dim=1)                                                   # (B,H,W)
    q = torch.quantile(energy.reshape(energy.shape[0], -1), q=quantile, dim=1, interpolation='nearest')[:, None, None]
    excess = energy - q
    mask = (excess > 0)[:, None, :, :]
    unit = grad / energy[:, None, :, :]
    unit[torch.isnan(unit)] = 0
    return torch.clamp(excess, min=0)[:, None, :, :] * unit, mask
