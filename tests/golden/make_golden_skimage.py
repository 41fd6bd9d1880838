#!/opt/conda/bin/python3.9
"""Fixture G20: the two scikit-image helpers of the reference's image2image path, computed by scikit-image ITSELF.

    /opt/conda/bin/python3.9 tests/golden/make_golden_skimage.py

The build container's system Python has no scikit-image; /opt/conda carries 0.18.3 for Python 3.9 (numpy 1.26, scipy 1.7 --
no torch there, so the reference module itself cannot be imported under that interpreter).  The reference pins 0.19.3
(requirements.txt); the three functions it calls did not change between the two releases except for keyword names
(`selem` still accepted in 0.19, `channel_axis=2` is 0.18's `multichannel=True`).  What runs here is the reference's own code
path, line by line, on numpy arrays:
  * dilate_mask (SinDDM/functions.py:21-33): morphology.disk(7 | 20) -> mask.permute(1,2,0)[:,:,0] -> morphology.binary_dilation(
    mask, selem=element) -> filters.gaussian(mask, sigma=5) -> (1,1,H,W) -> min-max normalisation;
  * match_histograms (SinDDM/trainer.py:312-314): exposure.match_histograms(image=uint8 HxWx3, reference=uint8 hxwx3, per channel).
Only inputs and outputs are written (tests/golden/g20_skimage.npz); tests/test_host.py holds sinddm_amd.functions to them."""
import os
import numpy as np
import skimage
from skimage import exposure, filters, morphology

HERE = os.path.dirname(os.path.abspath(__file__))


def dilate_mask_ref(mask_chw, mode):                       # SinDDM/functions.py:21-33 with numpy in place of torch
    if mode == "harmonization":
        element = morphology.disk(radius=7)
    if mode == "editing":
        element = morphology.disk(radius=20)
    mask = np.transpose(mask_chw, (1, 2, 0))               # mask.permute((1, 2, 0))
    mask = mask[:, :, 0]
    mask = morphology.binary_dilation(mask, selem=element)
    mask = filters.gaussian(mask, sigma=5)
    mask = mask[:, :, None, None]
    mask = mask.transpose(3, 2, 0, 1)
    mask = (mask - mask.min()) / (mask.max() - mask.min())
    return mask


def main():
    out = {"skimage_version": np.array(skimage.__version__)}
    rng = np.random.RandomState(20)
    masks = {}
    m = np.zeros((3, 94, 126), np.float32); m[:, 30:52, 40:75] = 1.0; masks["blob"] = m
    m = np.zeros((3, 94, 126), np.float32); m[:, 0:9, 0:14] = 1.0; m[:, 80:94, 110:126] = 1.0; m[:, 45, 60] = 1.0; masks["corners_and_a_pixel"] = m
    m = (rng.rand(1, 60, 80) > 0.995).astype(np.float32).repeat(3, 0); masks["speckles"] = m
    m = np.zeros((3, 60, 80), np.float32); m[0, 20:30, 20:30] = 0.4; m[1] = 1.0; masks["channel0_only_grey"] = m   # only channel 0 counts; 0.4 is "set"
    for name, mk in masks.items():
        out["mask_" + name] = mk
        for mode in ("harmonization", "editing"):
            out[f"dilate_{name}_{mode}"] = dilate_mask_ref(mk, mode)
    src = rng.randint(0, 256, size=(40, 50, 3)).astype(np.uint8)
    ref = (rng.randint(0, 128, size=(30, 20, 3)) + 64).astype(np.uint8)
    grad = np.stack([np.tile(np.arange(64, dtype=np.uint8) * 4, (48, 1))] * 3, -1)           # smooth ramp, few levels missing
    ref2 = rng.randint(0, 256, size=(33, 47, 3)).astype(np.uint8); ref2[..., 1] //= 3
    out.update(mh_src=src, mh_ref=ref, mh_out=exposure.match_histograms(image=src, reference=ref, multichannel=True),
               mh_src2=grad, mh_ref2=ref2, mh_out2=exposure.match_histograms(image=grad, reference=ref2, multichannel=True),
               mh_self=exposure.match_histograms(image=src, reference=src, multichannel=True))
    for k in ("mh_out", "mh_out2", "mh_self"):
        assert out[k].dtype == np.uint8, (k, out[k].dtype)
    np.savez_compressed(os.path.join(HERE, "g20_skimage.npz"), **out)
    print("wrote g20_skimage.npz", {k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    main()
