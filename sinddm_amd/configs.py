"""The five workload configurations of BASELINE.json made concrete.

`sizes` are (W, H) exactly as the reference's create_img_scales returns them for the named dataset
image / arguments and `rescale_losses` its wrapped-uint8 Frobenius losses; both were captured from the
reference by tests/golden/make_golden.py (tests/golden/g11_img_scales.json) and this table is checked
against that file by tests/test_host.py.  Plain data, so bench.py needs neither reference nor datasets.
"""
from __future__ import annotations

CONFIGS = {
    # balloons, image_size=(126,94), sf_in 1.411 -> 3 scales; T=100; B=1 (CPU plumbing config)
    "C1": dict(T=100, batch=1, scale_factor=1.399404635312222,
               sizes=[(64, 48), (90, 67), (126, 94)],
               rescale_losses=[1.0871835898797855, 0.7771932694518568],
               num_timesteps_ideal=[100, 52, 41]),
    # balloons, main.py defaults (auto_scale=50000) -> 5 scales; T=1000; B=16 on 1 GPU (headline metric config)
    "C2": dict(T=1000, batch=16, scale_factor=1.4030331316483415,
               sizes=[(64, 48), (90, 67), (126, 94), (177, 133), (248, 186)],
               rescale_losses=[1.0871835898797855, 0.7771932694518568, 0.5452509776707822, 0.3865868564044144],
               num_timesteps_ideal=[1000, 522, 416, 312, 228]),
    # seascape, image_size=(512,411), sf_in 1.5 -> 6 scales; T=1000; B=64 on 1 GPU
    "C3": dict(T=1000, batch=64, scale_factor=1.52396279130716,
               sizes=[(62, 50), (95, 76), (145, 116), (220, 177), (336, 270), (512, 411)],
               rescale_losses=[1.1605718897060657, 0.7589640417421595, 0.49976303898006297, 0.3227188177835286, 0.2005359996697877],
               num_timesteps_ideal=[1000, 543, 408, 289, 192, 119]),
    # starry_night, image_size=(252,198), sf_in 1.3 -> 6 scales; T=1000; B=128 over 8 GPUs (16/GPU)
    "C4": dict(T=1000, batch=128, scale_factor=1.3221898595574666,
               sizes=[(62, 49), (82, 65), (109, 86), (144, 113), (191, 150), (252, 198)],
               rescale_losses=[1.0107061373419781, 0.7626298107471386, 0.5817033421879454, 0.439163912219784, 0.3321110486601909],
               num_timesteps_ideal=[1000, 499, 410, 330, 257, 197]),
    # marinabaysands defaults, --scale_mul 2 4; T=1000; B=32 over 8 GPUs (4/GPU)
    "C5": dict(T=1000, batch=32, scale_factor=1.4103548263806815,
               sizes=[(69, 46), (97, 65), (137, 91), (194, 129), (273, 182)],
               rescale_losses=[1.1636340160921221, 0.803208026396207, 0.5658545551386978, 0.38791714917673686],
               num_timesteps_ideal=[1000, 544, 426, 322, 229],
               scale_mul=(2, 4)),
}


def build_diffusion(cfg_name: str, dim: int = 160, device=None, weights="closed_form"):
    """SinDDMNet + MultiScaleGaussianDiffusion for a named config with synthetic closed-form weights
    (published checkpoints are not available offline)."""
    from .models import MultiScaleGaussianDiffusion, SinDDMNet
    from .synth import closed_form_state_dict
    c = CONFIGS[cfg_name]
    net = SinDDMNet(dim=dim, multiscale=True, device=device)
    if device is not None:
        net.to(device)
    if weights == "closed_form":
        net.load_state_dict(closed_form_state_dict(dim))
    d = MultiScaleGaussianDiffusion(net, n_scales=len(c["sizes"]), scale_factor=c["scale_factor"],
                                    image_sizes=c["sizes"], scale_mul=c.get("scale_mul", (1, 1)), timesteps=c["T"],
                                    train_full_t=True, scale_losses=c["rescale_losses"], loss_factor=1,
                                    loss_type="l1", device=device, reblurring=True, omega=0)
    if device is not None:
        d.to(device)
    assert d.num_timesteps_ideal == c["num_timesteps_ideal"]
    return net, d
