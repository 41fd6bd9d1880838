"""tools/f44_model.py is the specification of conv_wino6.h (the experimental F(4x4,3x3) kernel): the 3 x 3 frequency-block
split over four waves, the two halves of the output transform and the device's operation order, in numpy.  It must stay a
convolution (reference SinDDM/models.py:63,65: nn.Conv2d(.., 3, padding=1) -- checked on the valid part)."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    spec = importlib.util.spec_from_file_location("f44_model", os.path.join(ROOT, "tools", "f44_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_block_split_is_exact_and_fp32_error_is_small():
    m = _model()
    rng = np.random.default_rng(3)
    x = rng.standard_normal((16, 10, 14))
    w = rng.uniform(-1, 1, (8, 16, 3, 3)) / 12
    got = m.conv_f44(x, w).astype(np.float64)
    ref = m.direct64(x, w)
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 5e-6


def test_transform_rows_match_the_matrices():
    m = _model()
    d = np.random.default_rng(4).standard_normal((6, 7))
    for a in range(2):
        assert np.allclose(m.bt3(d, a), m.BT[3 * a:3 * a + 3] @ d, atol=1e-5)
        mm = np.random.default_rng(5 + a).standard_normal((3, 7))
        assert np.allclose(m.at_rows(mm, a), m.AT[:, 3 * a:3 * a + 3] @ mm, atol=1e-5)
    t = np.random.default_rng(7).standard_normal((6, 5))
    assert np.allclose(m.at_cols(t), m.AT @ t, atol=1e-5)
