"""Host-side mirror of pytorch3d/renderer/compositing.py:19-247 over pytorch3d_amd._C.

Differences from the reference wrapper, all invisible to callers: no `.clone()` of the three
inputs for backward (compositing.py:50 -- the kernels never write their inputs) and no forced
contiguous copies of the permuted (N,H,W,K) views (the C ABI takes strides).
"""
import torch

from . import _C


def _make(forward_op, backward_op):
    class _Composite(torch.autograd.Function):
        @staticmethod
        def forward(ctx, features, alphas, points_idx):
            pt_cld = forward_op(features, alphas, points_idx)
            ctx.save_for_backward(features, alphas, points_idx)
            return pt_cld

        @staticmethod
        def backward(ctx, grad_output):
            features, alphas, points_idx = ctx.saved_tensors
            grad_features, grad_alphas = backward_op(grad_output, features, alphas, points_idx)
            return grad_features, grad_alphas, None

    return _Composite


_CompositeAlphaPoints = _make(_C.accum_alphacomposite, _C.accum_alphacomposite_backward)
_CompositeNormWeightedSumPoints = _make(_C.accum_weightedsumnorm, _C.accum_weightedsumnorm_backward)
_CompositeWeightedSumPoints = _make(_C.accum_weightedsum, _C.accum_weightedsum_backward)


def alpha_composite(pointsidx, alphas, pt_clds) -> torch.Tensor:
    """compositing.py:68-96.  pointsidx/alphas (N,K,H,W), pt_clds (C,P) -> (N,C,H,W)."""
    return _CompositeAlphaPoints.apply(pt_clds, alphas, pointsidx)


def norm_weighted_sum(pointsidx, alphas, pt_clds) -> torch.Tensor:
    """compositing.py:148-175."""
    return _CompositeNormWeightedSumPoints.apply(pt_clds, alphas, pointsidx)


def weighted_sum(pointsidx, alphas, pt_clds) -> torch.Tensor:
    """compositing.py:227-247."""
    return _CompositeWeightedSumPoints.apply(pt_clds, alphas, pointsidx)
