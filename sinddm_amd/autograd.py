"""torch.autograd glue: the training forward/backward of SinDDMNet and the L1 loss run in the HIP
library; autograd only carries the (B,3,H,W) gradient between them and exposes the parameter
gradients as the `.grad` views of the flat gradient buffer (so `loss.backward()` and optimizers keep
their usual semantics -- reference trainer.py:200-209, functions.py:97-102)."""
from __future__ import annotations

import torch

from . import _lib


# generation counter of the shared training workspace per device: a second training forward before the first
# backward() would silently overwrite the first graph's saved activations, so backward checks its stamp
_WS_GEN = {}


class _NetTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, net, t, scale):
        from .models import _workspace
        lib = _lib.load()
        x = x.contiguous()
        B, C, H, W = x.shape
        t = t.to(device=x.device, dtype=torch.int64).contiguous()
        out = torch.empty_like(x)
        packed = net.packed_weights()
        nbytes = lib.sinddm_train_workspace_bytes(net.dim, B, H, W)
        ws = _workspace(x.device, nbytes, tag="train")
        _lib.check(lib.sinddm_net_forward_train(_lib.ptr(net.flat_params), _lib.ptr(packed), _lib.ptr(x), _lib.ptr(t),
                                                0, float(scale), _lib.ptr(out), net.dim_arg, B, H, W, ws.data_ptr(),
                                                ws.numel(), _lib.stream_ptr(x.device)), "sinddm_net_forward_train")
        key = ws.data_ptr()
        _WS_GEN[key] = _WS_GEN.get(key, 0) + 1
        ctx.ws_gen = (key, _WS_GEN[key])
        ctx.net, ctx.ws, ctx.shape = net, ws, (B, H, W)
        ctx.save_for_backward(x)
        ctx.need_gx = x.requires_grad
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        net = ctx.net
        (x,) = ctx.saved_tensors
        B, H, W = ctx.shape
        key, gen = ctx.ws_gen
        if ctx.ws.data_ptr() != key or _WS_GEN.get(key) != gen:
            raise _lib.SinddmError("the saved activations of this graph were overwritten by a later training forward "
                                   "(one live SinDDMNet graph per device: call backward() before the next forward)")
        grad_out = grad_out.contiguous()
        gx = torch.empty_like(x) if ctx.need_gx else None
        packed = net.packed_weights()
        packed_bwd = net.packed_weights_bwd()
        net.bind_grads()
        _lib.check(lib.sinddm_net_backward(_lib.ptr(net.flat_params), _lib.ptr(packed), _lib.ptr(packed_bwd),
                                           _lib.ptr(x), _lib.ptr(grad_out), _lib.ptr(net.flat_grads),
                                           _lib.ptr(gx) if gx is not None else None, net.dim_arg, B, H, W,
                                           ctx.ws.data_ptr(), ctx.ws.numel(), _lib.stream_ptr(x.device)),
                   "sinddm_net_backward")
        return gx, None, None, None, None


def net_forward_train(net, x, t, scale):
    """Forward that saves activations; parameter grads are accumulated straight into net.flat_grads by
    the backward kernels.  `anchor` is a dummy differentiable input so autograd schedules backward()
    even when x itself does not require grad (the usual training case)."""
    if not x.is_cuda:
        raise _lib.SinddmError("SinDDMNet needs a ROCm device tensor: there is no CPU fallback")
    anchor = net._autograd_anchor()
    return _NetTrainFn.apply(x, anchor, net, t, scale)


class _L1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, noise, eps):
        lib = _lib.load()
        noise = noise.contiguous()
        eps = eps.contiguous()
        loss = torch.zeros((), dtype=torch.float32, device=eps.device)
        g = torch.empty_like(eps)
        _lib.check(lib.sinddm_l1_loss_fwd_bwd(_lib.ptr(noise), _lib.ptr(eps), _lib.ptr(loss), _lib.ptr(g), eps.numel(),
                                              1.0, _lib.stream_ptr(eps.device)), "sinddm_l1_loss_fwd_bwd")
        ctx.save_for_backward(g)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        (g,) = ctx.saved_tensors
        # d loss / d eps = -sign(noise - eps) / N, scaled by the upstream scalar (e.g. 1/grad_accumulate)
        return None, g * grad_loss


def l1_loss(noise, eps):
    """mean(|noise - eps|)  (reference models.py:594)."""
    return _L1Fn.apply(noise, eps)
