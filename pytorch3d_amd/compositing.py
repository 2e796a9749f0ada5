"""Point compositing on the HIP kernels: `alpha_composite`, `norm_weighted_sum`, `weighted_sum`.

Same call signatures as the reference (pytorch3d/renderer/compositing.py:68-96, 148-175, 227-247): indices and alphas of
shape (N, K, H, W), features (C, P), result (N, C, H, W).  One autograd node serves the three modes; it keeps references
to its inputs (the kernels never write them, so the reference's three `.clone()`s are not needed) and hands the
renderer's permuted (N, H, W, K) views to the C ABI as they are (it takes strides, no `.contiguous()` copies).
"""
import torch

from . import _C

_KERNELS = {
    "alpha": (_C.accum_alphacomposite, _C.accum_alphacomposite_backward),
    "norm": (_C.accum_weightedsumnorm, _C.accum_weightedsumnorm_backward),
    "sum": (_C.accum_weightedsum, _C.accum_weightedsum_backward),
}


class _Compose(torch.autograd.Function):
    """(mode, features (C,P), alphas (N,K,H,W), point indices (N,K,H,W)) -> images (N,C,H,W); gradients to the features
    and the alphas."""

    @staticmethod
    def forward(ctx, mode, features, alphas, indices):
        ctx.mode = mode
        ctx.save_for_backward(features, alphas, indices)
        return _KERNELS[mode][0](features, alphas, indices)

    @staticmethod
    def backward(ctx, grad_images):
        features, alphas, indices = ctx.saved_tensors
        g_features, g_alphas = _KERNELS[ctx.mode][1](grad_images, features, alphas, indices)
        return None, g_features, g_alphas, None


def alpha_composite(pointsidx, alphas, pt_clds) -> torch.Tensor:
    """Front-to-back alpha compositing: sum_k f[idx_k] * a_k * prod_{l<k} (1 - a_l)."""
    return _Compose.apply("alpha", pt_clds, alphas, pointsidx)


def norm_weighted_sum(pointsidx, alphas, pt_clds) -> torch.Tensor:
    """sum_k a_k f[idx_k] / max(sum_k a_k, 1e-4)."""
    return _Compose.apply("norm", pt_clds, alphas, pointsidx)


def weighted_sum(pointsidx, alphas, pt_clds) -> torch.Tensor:
    """sum_k a_k f[idx_k]."""
    return _Compose.apply("sum", pt_clds, alphas, pointsidx)
