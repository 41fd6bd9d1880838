"""Deterministic, RNG-free synthetic weights and inputs.

Published SinDDM checkpoints are not available offline, so parity fixtures, the
smoke test and the benchmark all use the same closed-form weight fill: every
tensor k of the state dict gets  w.flat[i] = a_k * sin(0.37*i + k)  with a_k
~ 1/sqrt(fan_in), so activations stay O(1) through the four conv blocks.
The same function fills the reference model (tests/golden/make_golden.py) and
the HIP-backed model, so fixtures hold inputs/outputs only.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch


def net_param_shapes(dim: int = 160, channels: int = 3, time_dim: int = 32) -> "OrderedDict[str, Tuple[int, ...]]":
    """State-dict keys and shapes of SinDDMNet(dim, multiscale=True), in
    nn.Module registration order (reference SinDDM/models.py:54-67,106-110,124-132)."""
    half = int(dim / 2)
    shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    shapes["time_mlp.0.weight"] = (time_dim * 4, time_dim * 2)
    shapes["time_mlp.0.bias"] = (time_dim * 4,)
    shapes["time_mlp.2.weight"] = (time_dim, time_dim * 4)
    shapes["time_mlp.2.bias"] = (time_dim,)
    for name, (cin, cout) in zip(("l1", "l2", "l3", "l4"),
                                 ((channels, half), (half, dim), (dim, dim), (dim, half))):
        shapes[f"{name}.mlp.1.weight"] = (time_dim, time_dim)
        shapes[f"{name}.mlp.1.bias"] = (time_dim,)
        shapes[f"{name}.time_reshape.weight"] = (cin, time_dim, 1, 1)
        shapes[f"{name}.time_reshape.bias"] = (cin,)
        shapes[f"{name}.ds_conv.weight"] = (cin, 1, 5, 5)
        shapes[f"{name}.ds_conv.bias"] = (cin,)
        shapes[f"{name}.net.0.weight"] = (cout, cin, 3, 3)
        shapes[f"{name}.net.0.bias"] = (cout,)
        shapes[f"{name}.net.2.weight"] = (cout, cout, 3, 3)
        shapes[f"{name}.net.2.bias"] = (cout,)
        if cin != cout:
            shapes[f"{name}.res_conv.weight"] = (cout, cin, 1, 1)
            shapes[f"{name}.res_conv.bias"] = (cout,)
    shapes["final_conv.0.weight"] = (channels, half, 1, 1)
    shapes["final_conv.0.bias"] = (channels,)
    return shapes


def closed_form_state_dict(dim: int = 160, channels: int = 3, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Closed-form fill described in the module docstring (float64 sin, cast to f32)."""
    sd: Dict[str, torch.Tensor] = OrderedDict()
    for k, (name, shape) in enumerate(net_param_shapes(dim, channels).items()):
        n = int(np.prod(shape))
        if name.endswith("bias"):
            amp = 0.05
        else:
            fan_in = int(np.prod(shape[1:]))
            amp = gain * math.sqrt(2.0 / fan_in)
        i = np.arange(n, dtype=np.float64)
        w = amp * np.sin(0.37 * i + k)
        sd[name] = torch.tensor(w.reshape(shape), dtype=torch.float32)
    return sd


def closed_form_tensor(shape, phase: float = 0.0, amp: float = 1.0, freq: float = 0.618) -> torch.Tensor:
    """RNG-free test input: amp * sin(freq*i + phase) reshaped."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    return torch.tensor((amp * np.sin(freq * i + phase)).reshape(shape), dtype=torch.float32)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_randn(shape, key: int) -> torch.Tensor:
    """Counter-based N(0,1) draws that do not depend on any library RNG state:
    splitmix64(index, key) -> two uniforms -> Box-Muller in float64 -> f32.
    Used wherever a parity test needs 'recorded noise' without storing it."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        k = np.uint64((int(key) * 0x2545F4914F6CDD1D + 0x1234567) & 0xFFFFFFFFFFFFFFFF)
        h1 = _splitmix64(idx * np.uint64(2) + k)
        h2 = _splitmix64(idx * np.uint64(2) + np.uint64(1) + k)
    u1 = ((h1 >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)
    u2 = ((h2 >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return torch.tensor(z.reshape(shape), dtype=torch.float32)


def noise_key(kind: str, s: int = 0, t: int = 0, rank: int = 0) -> int:
    """Stable integer key for the three kinds of draws of the sampler
    ('init', 'renoise', 'step'), so a chain can be replayed anywhere."""
    base = {"init": 1, "renoise": 2, "step": 3, "train": 4}[kind]
    return ((rank * 7 + base) * 64 + int(s)) * 100003 + int(t)
