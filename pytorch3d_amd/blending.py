"""Host-side mirror of pytorch3d/renderer/blending.py:43-244 over the C ABI.

`hard_rgb_blend` (blending.py:54-88) is one kernel each way (p3d_hard_rgb_blend_forward / _backward).

`sigmoid_alpha_blend` goes through `pytorch3d_amd._C.sigmoid_alpha_blend[_backward]` exactly like the
reference's wrapper (blending.py:95-140).  `softmax_rgb_blend`, ~20 elementwise torch ops plus their autograd
graph in the reference (blending.py:147-244), is ONE kernel forward and ONE backward here
(include/p3d_amd.h: p3d_softmax_rgb_blend_forward / _backward); same arguments, defaults and return value.
"""
import ctypes
from typing import NamedTuple, Sequence, Union

import torch

from . import _C, _lib


class BlendParams(NamedTuple):
    """blending.py:20-40."""

    sigma: float = 1e-4
    gamma: float = 1e-4
    background_color: Union[torch.Tensor, Sequence[float]] = (1.0, 1.0, 1.0)


class _SigmoidAlphaBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dists, pix_to_face, sigma):
        alphas = _C.sigmoid_alpha_blend(dists, pix_to_face, sigma)
        ctx.save_for_backward(dists, pix_to_face, alphas)
        ctx.sigma = sigma
        return alphas

    @staticmethod
    def backward(ctx, grad_alphas):
        dists, pix_to_face, alphas = ctx.saved_tensors
        grad_dists = _C.sigmoid_alpha_blend_backward(grad_alphas, alphas, dists, pix_to_face, ctx.sigma)
        return grad_dists, None, None


def sigmoid_alpha_blend(colors, fragments, blend_params: BlendParams) -> torch.Tensor:
    """blending.py:118-144: RGB of the closest face, alpha from the 2D distance probability map."""
    N, H, W, K = fragments.pix_to_face.shape
    pixel_colors = torch.ones((N, H, W, 4), dtype=colors.dtype, device=colors.device)
    pixel_colors[..., :3] = colors[..., 0, :]
    pixel_colors[..., 3] = _SigmoidAlphaBlend.apply(fragments.dists, fragments.pix_to_face, blend_params.sigma)
    return pixel_colors


def _background(blend_params, device, who="softmax_rgb_blend"):
    bg = blend_params.background_color
    if isinstance(bg, torch.Tensor):
        if bg.requires_grad:
            # the reference keeps the background in the autograd graph (blending.py:183-186); the fused kernel takes it
            # as three constants -- refuse rather than return a silently missing gradient
            raise NotImplementedError(who + ": a background_color that requires grad is not supported by the "
                                      "fused kernel (it is passed as constants); detach it or blend with torch ops")
        bg = [float(x) for x in bg.detach().reshape(-1).tolist()]
    bg = [float(x) for x in bg]
    if len(bg) != 3:
        raise ValueError("background_color must have 3 elements")
    return (ctypes.c_float * 3)(*bg)


def _plane(v, N, device):
    """znear / zfar: float, or a per-batch-element tensor (blending.py:205-210).  -> (scalar, device tensor or None)"""
    if torch.is_tensor(v):
        if v.requires_grad:
            raise NotImplementedError("softmax_rgb_blend: znear / zfar tensors that require grad are not supported by the "
                                      "fused kernel; detach them")
        t = v.detach().to(device=device, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(N)
        if t.numel() != N:
            raise ValueError("znear / zfar tensors must have one entry per batch element")
        return 0.0, t.contiguous()
    return float(v), None


class _SoftmaxRGBBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, colors, dists, zbuf, pix_to_face, sigma, gamma, bg, znear, zfar):
        N, H, W, K = pix_to_face.shape
        dev = colors.device
        c = colors.contiguous()
        d, z, p2f = dists.contiguous(), zbuf.contiguous(), pix_to_face.contiguous()
        zn, zn_t = _plane(znear, N, dev)
        zf, zf_t = _plane(zfar, N, dev)
        lib = _lib.load()
        with torch.cuda.device(dev):
            out = torch.empty((N, H, W, 4), dtype=torch.float32, device=dev)
            if out.numel():
                rc = lib.p3d_softmax_rgb_blend_forward(_C._ptr(c), _C._ptr(p2f), _C._ptr(d), _C._ptr(z), float(sigma),
                                                       float(gamma), bg, zn, zf, _C._ptr(zn_t), _C._ptr(zf_t), N, H * W,
                                                       K, _C._ptr(out), _C._stream(dev))
                _lib.check(rc, "softmax_rgb_blend")
        ctx.save_for_backward(c, d, z, p2f, zn_t if zn_t is not None else torch.empty(0, device=dev),
                              zf_t if zf_t is not None else torch.empty(0, device=dev))
        ctx.params = (float(sigma), float(gamma), bg, zn, zf, zn_t is not None, zf_t is not None)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        c, d, z, p2f, zn_t, zf_t = ctx.saved_tensors
        sigma, gamma, bg, zn, zf, has_zn, has_zf = ctx.params
        N, H, W, K = p2f.shape
        dev = c.device
        g = grad_out.contiguous()
        lib = _lib.load()
        with torch.cuda.device(dev):
            gc = torch.empty((N, H, W, K, 3), dtype=torch.float32, device=dev)
            gd = torch.empty((N, H, W, K), dtype=torch.float32, device=dev)
            gz = torch.empty((N, H, W, K), dtype=torch.float32, device=dev)
            if gd.numel():
                rc = lib.p3d_softmax_rgb_blend_backward(_C._ptr(g), _C._ptr(c), _C._ptr(p2f), _C._ptr(d), _C._ptr(z),
                                                        sigma, gamma, bg, zn, zf, _C._ptr(zn_t if has_zn else None),
                                                        _C._ptr(zf_t if has_zf else None), N, H * W, K, _C._ptr(gc),
                                                        _C._ptr(gd), _C._ptr(gz), _C._stream(dev))
                _lib.check(rc, "softmax_rgb_blend_backward")
        return gc, gd, gz, None, None, None, None, None, None


def softmax_rgb_blend(colors, fragments, blend_params: BlendParams, znear: Union[float, torch.Tensor] = 1.0,
                      zfar: Union[float, torch.Tensor] = 100) -> torch.Tensor:
    """blending.py:147-244.  colors (N,H,W,K,3), fragments.{pix_to_face, dists, zbuf} (N,H,W,K) -> RGBA (N,H,W,4)."""
    _C._check_fragments("softmax_rgb_blend", fragments.pix_to_face, colors=colors, dists=fragments.dists, zbuf=fragments.zbuf)
    bg = _background(blend_params, colors.device)
    return _SoftmaxRGBBlend.apply(colors, fragments.dists, fragments.zbuf, fragments.pix_to_face, blend_params.sigma,
                                  blend_params.gamma, bg, znear, zfar)


class _HardRGBBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, colors, pix_to_face, bg):
        N, H, W, K = pix_to_face.shape
        dev = colors.device
        c, p2f = colors.contiguous(), pix_to_face.contiguous()
        lib = _lib.load()
        with torch.cuda.device(dev):
            out = torch.empty((N, H, W, 4), dtype=torch.float32, device=dev)
            if out.numel():
                rc = lib.p3d_hard_rgb_blend_forward(_C._ptr(c), _C._ptr(p2f), bg, N * H * W, K, _C._ptr(out), _C._stream(dev))
                _lib.check(rc, "hard_rgb_blend")
        ctx.save_for_backward(p2f)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (p2f,) = ctx.saved_tensors
        N, H, W, K = p2f.shape
        dev = p2f.device
        g = grad_out.contiguous()
        lib = _lib.load()
        with torch.cuda.device(dev):
            gc = torch.empty((N, H, W, K, 3), dtype=torch.float32, device=dev)
            if gc.numel():
                rc = lib.p3d_hard_rgb_blend_backward(_C._ptr(g), _C._ptr(p2f), N * H * W, K, _C._ptr(gc), _C._stream(dev))
                _lib.check(rc, "hard_rgb_blend_backward")
        return gc, None, None


def hard_rgb_blend(colors, fragments, blend_params: BlendParams) -> torch.Tensor:
    """blending.py:54-88: RGB of the closest face (slot 0), the background colour where no face covers the pixel;
    alpha 1 / 0.  colors (N,H,W,K,3) -> RGBA (N,H,W,4)."""
    _C._check_fragments("hard_rgb_blend", fragments.pix_to_face, colors=colors)
    if colors.shape != tuple(fragments.pix_to_face.shape) + (3,) or fragments.pix_to_face.shape[3] < 1:
        raise ValueError("colors must have shape (N, H, W, K, 3) with K >= 1 matching pix_to_face")
    return _HardRGBBlend.apply(colors, fragments.pix_to_face, _background(blend_params, colors.device, "hard_rgb_blend"))
