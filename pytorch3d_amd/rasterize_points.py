"""Host-side mirror of pytorch3d/renderer/points/rasterize_points.py:24-242 over pytorch3d_amd._C."""
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from . import _C
from .rasterize_meshes import parse_image_size

kMaxPointsPerBin = 22  # rasterize_points.py:21


def rasterize_points(
    pointclouds,
    image_size: Union[int, List[int], Tuple[int, int]] = 256,
    radius: Union[float, List, Tuple, torch.Tensor] = 0.01,
    points_per_pixel: int = 8,
    bin_size: Optional[int] = None,
    max_points_per_bin: Optional[int] = None,
):
    """Returns (idx int32, zbuf, dists2), each (N, H, W, points_per_pixel)."""
    points_packed = pointclouds.points_packed()
    cloud_to_packed_first_idx = pointclouds.cloud_to_packed_first_idx()
    num_points_per_cloud = pointclouds.num_points_per_cloud()
    radius = _format_radius(radius, pointclouds)
    im_size = parse_image_size(image_size)
    max_image_size = max(*im_size)
    if bin_size is None:
        # rasterize_points.py:104-113 -- unlike meshes there is no "<= 64 -> 8" special case
        bin_size = int(2 ** max(np.ceil(np.log2(max_image_size)) - 4, 4))
    if bin_size != 0:
        points_per_bin = 1 + (max_image_size - 1) // bin_size
        if points_per_bin >= kMaxPointsPerBin:
            raise ValueError("bin_size too small, number of points per bin must be less than %d; got %d" %
                             (kMaxPointsPerBin, points_per_bin))
    if max_points_per_bin is None:
        max_points_per_bin = int(max(10000, pointclouds._P / 5))
    return _RasterizePoints.apply(points_packed, cloud_to_packed_first_idx, num_points_per_cloud, im_size, radius,
                                  points_per_pixel, bin_size, max_points_per_bin)


def _format_radius(radius, pointclouds) -> torch.Tensor:
    """rasterize_points.py:145-184: float / list / tuple / (N, P_padded) tensor -> (P_packed,)."""
    N, P_padded = pointclouds._N, pointclouds._P
    points_packed = pointclouds.points_packed()
    P_packed = points_packed.shape[0]
    if isinstance(radius, (list, tuple)):
        radius = torch.tensor(radius).type_as(points_packed)
    if isinstance(radius, torch.Tensor):
        if N == 1 and radius.ndim == 1:
            radius = radius[None, ...]
        if radius.shape != (N, P_padded):
            raise ValueError("radius must be of shape (N, P): got %s" % repr(radius.shape))
        radius = radius.view(-1)[pointclouds.padded_to_packed_idx()]
    elif isinstance(radius, float):
        # same values as the reference's CPU-side fill + type_as, without the host buffer and the H2D copy
        radius = torch.full((P_packed,), radius, dtype=points_packed.dtype, device=points_packed.device)
    else:
        raise ValueError("radius must be a float, list, tuple or tensor; got %s" % type(radius))
    return radius


class _RasterizePoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, cloud_to_packed_first_idx, num_points_per_cloud, image_size=(256, 256), radius=0.01,
                points_per_pixel=8, bin_size=0, max_points_per_bin=0):
        idx, zbuf, dists = _C.rasterize_points(points, cloud_to_packed_first_idx, num_points_per_cloud, image_size,
                                               radius, points_per_pixel, bin_size, max_points_per_bin)
        ctx.save_for_backward(points, idx)
        ctx.mark_non_differentiable(idx)
        return idx, zbuf, dists

    @staticmethod
    def backward(ctx, grad_idx, grad_zbuf, grad_dists):
        points, idx = ctx.saved_tensors
        grad_points = _C.rasterize_points_backward(points, idx, grad_zbuf, grad_dists)
        return (grad_points,) + (None,) * 7
