"""tools/h2_model.py is the numerical specification of the binary16 hi/lo 3x3 kernels (conv_h2.h, conv_wh.h): the split, the
three / four MFMA terms with fp32 accumulation, the exact power-of-two scales, the Winograd transforms in the kernel's
order.  CPU checks: the split keeps 22 significant bits, both schemes are convolutions, and on network-shaped data their
error against float64 is not above the error of an fp32 FMA chain over the same sum (the gate the GPU test repeats on
hardware: tests/test_gpu_h2.py).                           reference SinDDM/models.py:63,65 (the 3x3 convolutions)"""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    spec = importlib.util.spec_from_file_location("h2_model", os.path.join(ROOT, "tools", "h2_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_split_keeps_22_bits_over_the_dynamic_range():
    m = _model()
    rng = np.random.default_rng(1)
    a = (rng.standard_normal(20000) * np.exp(rng.uniform(-6, 0, 20000))).astype(np.float32)      # magnitudes over ~2.6 decades
    for target in (13, 10):
        sh = m.shift_for(np.abs(a).max(), target)
        hi, lo = m.split(a, sh)
        assert np.abs(hi).max() < 65504 and np.all(np.isfinite(hi)) and np.all(np.isfinite(lo))
        back = (hi + lo) * 2.0 ** -sh
        err = np.abs(back - a.astype(np.float64))
        # 2^-22 of the element, or the binary16 subnormal floor 2^-25 of the scaled unit for the small ones
        assert np.all(err <= np.maximum(2.0 ** -22 * np.abs(a), 2.0 ** -25 * 2.0 ** -sh) * 1.0001)


def test_both_schemes_are_the_convolution_and_not_wider_than_fp32():
    m = _model()
    x, w = m.network_like(48, 16, 8, 16, seed=3)
    ref = m.conv_direct64(x, w)
    e32 = m.rel(m.conv_fp32(x, w), ref)
    e_h2 = m.rel(m.conv_h2(x, w), ref)
    e_wh = m.rel(m.conv_wh(x, w), ref)
    assert e_h2 < 1e-6 and e_wh < 1e-6                      # they ARE the convolution
    assert e_h2 <= 1.5 * e32 and e_wh <= 1.5 * e32, (e_h2, e_wh, e32)


def test_dynamic_range_of_the_scales():
    """Tensors far from 1 (the running-max scale) and weight rows of very different magnitude (the per-channel scale)."""
    m = _model()
    x, w = m.network_like(32, 8, 4, 8, seed=5)
    ref = m.conv_direct64(x, w)
    for gx, gw in ((2.0 ** -14, 1.0), (2.0 ** 12, 1.0), (1.0, 2.0 ** -20), (3.1e3, 7.7e-5)):
        xs, wsc = (x * np.float32(gx)).astype(np.float32), w.copy()
        wsc[::2] *= np.float32(gw)                            # every other output channel rescaled
        r = m.conv_direct64(xs, wsc)
        for fn in (m.conv_h2, m.conv_wh):
            assert m.rel(fn(xs, wsc), r) < 1e-6, (fn.__name__, gx, gw)
