"""PointsRenderer's chain -- rasterize_points, weights = 1 - dists / r^2, alpha_composite (or norm_weighted_sum) -- as ONE autograd
node on two launches.

The reference (pytorch3d/renderer/points/renderer.py:56-76) runs the rasterizer, two element-wise kernels, two permuted copies and the
compositor, and in the backward the compositor's backward, the element-wise backwards and the rasterizer's backward.  Here the pixel of
the image is formed in the fine kernel's epilogue (include/p3d_amd.h: p3d_rasterize_points_composite) and the backward is one kernel
whose two scatters share a table (p3d_rasterize_points_composite_backward).  Same bits in the image as the operators one after the
other; gradients within the compositor's and the rasterizer's own tolerances (tests/test_gpu_render_points.py).

`render_points_alpha` takes what `rasterize_points` takes plus the packed features; pytorch3d_amd.shim patches PointsRenderer.forward
onto it when the renderer is the plain (PointsRasterizer, AlphaCompositor) pair.
"""
from typing import Optional, Sequence, Tuple, Union

import torch

from . import _C
from .rasterize_meshes import parse_image_size
from .rasterize_points import MAX_BINS_PER_SIDE, default_bin_size, radius_per_packed_point

MAX_FUSED_K = 16  # include/p3d_amd.h: p3d_rasterize_points_composite_backward
MAX_FUSED_C = 4


def fusable(features_packed, radius, points_per_pixel) -> bool:
    """Can this chain run as the fused node?  (float32 (P, C <= 4) features on the GPU, one scalar radius, K <= 16, the default tie
    order.)  Otherwise: the operators one after the other."""
    return (torch.is_tensor(features_packed) and features_packed.is_cuda and features_packed.dtype == torch.float32
            and features_packed.dim() == 2 and 1 <= features_packed.shape[1] <= MAX_FUSED_C
            and isinstance(radius, (float, int)) and not isinstance(radius, bool) and float(radius) > 0.0
            and 0 < int(points_per_pixel) <= MAX_FUSED_K and not _C.CUDA_TIE_ORDER)


def render_points_alpha(pointclouds, features_packed, image_size: Union[int, Sequence[int]] = 256, radius: float = 0.01,
                        points_per_pixel: int = 8, bin_size: Optional[int] = None, max_points_per_bin: Optional[int] = None,
                        weight_radius: Optional[float] = None, radius_per_point: Optional[torch.Tensor] = None,
                        compositor: str = "alpha") -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """(images (N, H, W, C), idx int32, zbuf, dists2) of `pointclouds` (NDC points; the accessors of pytorch3d_amd.structures.
    PackedPointclouds) with `features_packed` (P, C).  image_size .. max_points_per_bin as rasterize_points (rasterize_points.py:24-132:
    same heuristics, same error).  weight_radius: the r of `weights = 1 - dists / r^2` when it is not the rasterization radius
    (PointsRenderer reads it from the rasterizer's own settings, renderer.py:62).  Gradients reach the points (x, y: the chain does
    not use zbuf) and the features.  compositor: "alpha" (AlphaCompositor, compositing.py:68-123) or "norm" (NormWeightedCompositor:
    norm_weighted_sum, compositing.py:126-185)."""
    if compositor not in ("alpha", "norm"):
        raise ValueError("compositor must be 'alpha' or 'norm'")
    size = parse_image_size(image_size)
    longest = max(size)
    if bin_size is None:
        bin_size = default_bin_size(longest)
    if bin_size != 0:
        bins = 1 + (longest - 1) // bin_size
        if bins >= MAX_BINS_PER_SIDE:
            raise ValueError("bin_size too small, number of points per bin must be less than %d; got %d" % (MAX_BINS_PER_SIDE, bins))
    if max_points_per_bin is None:
        max_points_per_bin = max(10000, pointclouds._P // 5)  # rasterize_points.py:125
    if not fusable(features_packed, radius if weight_radius is None else weight_radius, points_per_pixel):
        raise ValueError("render_points_alpha: float32 (P, C <= 4) GPU features, a scalar radius and points_per_pixel <= 16")
    rad = radius_per_point if radius_per_point is not None else radius_per_packed_point(radius, pointclouds)
    inv_r2 = _C.inv_r2_of(radius if weight_radius is None else weight_radius)
    return _SplatAlpha.apply(pointclouds.points_packed(), features_packed, rad, pointclouds.cloud_to_packed_first_idx(),
                             pointclouds.num_points_per_cloud(),
                             (size, int(points_per_pixel), int(bin_size), int(max_points_per_bin), inv_r2, compositor))


def render_points(*args, **kwargs):
    """render_points_alpha under its general name (the compositor is an argument)."""
    return render_points_alpha(*args, **kwargs)


class _SplatAlpha(torch.autograd.Function):
    """(points, features, radius, first, count, static) -> (images, idx, zbuf, dists2); gradients to the points and the features
    through `images` only (the fragments come along for callers that want to look at them)."""

    @staticmethod
    def forward(ctx, points, features, radius, first, count, static):
        size, k, bin_size, cap, inv_r2, mode = static
        idx, zbuf, dists2, images = _C.rasterize_points_composite(points, first, count, size, radius, features, inv_r2, k, bin_size, cap,
                                                                  mode)
        ctx.mode = mode
        ctx.save_for_backward(points, features, idx, dists2)
        ctx.mark_non_differentiable(idx, zbuf, dists2)
        ctx.set_materialize_grads(False)
        ctx.inv_r2 = inv_r2
        return images, idx, zbuf, dists2

    @staticmethod
    def backward(ctx, g_images, _g_idx, _g_zbuf, _g_dists2):
        if g_images is None:
            return (None,) * 6
        points, features, idx, dists2 = ctx.saved_tensors
        gp, gf = _C.rasterize_points_composite_backward(points, features, idx, dists2, g_images.contiguous(), ctx.inv_r2, ctx.mode)
        return (gp if ctx.needs_input_grad[0] else None), (gf if ctx.needs_input_grad[1] else None), None, None, None, None
