"""Fused Adam and EMA over the flat parameter buffer of SinDDMNet (one kernel launch each).

Replaces torch.optim.Adam(params, lr) (reference trainer.py:134,208 -- torch defaults: betas
(0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) and EMA.update_model_average
(models.py:23-31, trainer.py:155-159), which in eager mode loop over 52 tensors.
"""
from __future__ import annotations

import math

import torch

from . import _lib

MODE_ADAM, MODE_EMA_COPY, MODE_EMA_LERP = 1, 2, 4


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Optimizer-compatible wrapper (param_groups / lr schedulers / state_dict work) whose
    step() is one HIP kernel over the flat buffers."""

    def __init__(self, net, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.net = net
        super().__init__(list(net.parameters()), dict(lr=lr, betas=betas, eps=eps))
        self._step = 0
        self.exp_avg = torch.zeros_like(net.flat_params)
        self.exp_avg_sq = torch.zeros_like(net.flat_params)
        net.bind_grads()

    @torch.no_grad()
    def step(self, closure=None):
        lib = _lib.load()
        net = self.net
        if self.exp_avg.device != net.flat_params.device:
            self.exp_avg = self.exp_avg.to(net.flat_params.device)
            self.exp_avg_sq = self.exp_avg_sq.to(net.flat_params.device)
        g = self.param_groups[0]
        b1, b2 = g['betas']
        self._step += 1
        bc1 = 1 - b1 ** self._step
        bc2 = 1 - b2 ** self._step
        _lib.check(lib.sinddm_adam_ema_step(
            _lib.ptr(net.flat_params), _lib.ptr(net.flat_grads), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
            None, g['lr'] / bc1, b1, b2, g['eps'], math.sqrt(bc2), 0.0, 0.0, MODE_ADAM, net.flat_params.numel(),
            _lib.stream_ptr(net.flat_params.device)), "sinddm_adam_ema_step")
        net.mark_dirty()

    def zero_grad(self, set_to_none: bool = False):
        self.net.flat_grads.zero_()
        self.net.bind_grads()

    def state_dict(self):
        sd = super().state_dict()
        sd['fused'] = dict(step=self._step, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq)
        return sd

    def load_state_dict(self, sd):
        fused = sd.get('fused')
        super().load_state_dict({k: v for k, v in sd.items() if k != 'fused'})
        if fused is not None:
            self._step = int(fused['step'])
            self.exp_avg.copy_(fused['exp_avg'])
            self.exp_avg_sq.copy_(fused['exp_avg_sq'])


@torch.no_grad()
def ema_update_(ema_net, net, decay: float) -> None:
    """ema = decay*ema + (1-decay)*p over the flat buffers (models.py:28-31)."""
    lib = _lib.load()
    _lib.check(lib.sinddm_adam_ema_step(
        _lib.ptr(net.flat_params), None, None, None, _lib.ptr(ema_net.flat_params), 0.0, 0.0, 0.0, 0.0, 1.0,
        float(decay), 0.0, MODE_EMA_LERP, net.flat_params.numel(), _lib.stream_ptr(net.flat_params.device)),
        "sinddm_adam_ema_step")
    ema_net.mark_dirty()


@torch.no_grad()
def ema_copy_(ema_net, net) -> None:
    """ema = p over the flat buffers: what `EMA.reset_parameters` / `step_ema` before step_start_ema do with
    load_state_dict over 52 + 13 tensors (reference trainer.py:152-157) as ONE launch (mode 2 of sinddm_adam_ema_step)."""
    lib = _lib.load()
    _lib.check(lib.sinddm_adam_ema_step(
        _lib.ptr(net.flat_params), None, None, None, _lib.ptr(ema_net.flat_params), 0.0, 0.0, 0.0, 0.0, 1.0,
        0.0, 0.0, MODE_EMA_COPY, net.flat_params.numel(), _lib.stream_ptr(net.flat_params.device)),
        "sinddm_adam_ema_step")
    ema_net.mark_dirty()
