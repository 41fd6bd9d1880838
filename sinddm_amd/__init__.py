"""sinddm_amd -- MI355X (gfx950) native implementation of the SinDDM multi-scale diffusion hot path.

Public surface mirrors the reference package (`SinDDM.functions`, `SinDDM.models`, `SinDDM.trainer`):
    from sinddm_amd import SinDDMNet, MultiScaleGaussianDiffusion, MultiscaleTrainer, create_img_scales
Heavy imports are lazy so that host-only helpers work without the HIP library.
"""
__all__ = ["SinDDMNet", "MultiScaleGaussianDiffusion", "MultiscaleTrainer", "create_img_scales", "EMA"]


def __getattr__(name):
    if name in ("SinDDMNet", "MultiScaleGaussianDiffusion", "EMA", "SinusoidalPosEmb"):
        from . import models
        return getattr(models, name)
    if name in ("MultiscaleTrainer", "Dataset"):
        from . import trainer
        return getattr(trainer, name)
    if name == "create_img_scales":
        from .functions import create_img_scales
        return create_img_scales
    raise AttributeError(name)
