"""Point-cloud rasterization over `PackedPointclouds` (pytorch3d_amd/structures.py) on the HIP kernels.

Same call as the reference's `rasterize_points` (pytorch3d/renderer/points/rasterize_points.py:24-132: argument names,
defaults, the returned `(idx int32, zbuf, dists2)` of shape (N, H, W, points_per_pixel), the bin-size heuristic and its
error), so code written against the reference keeps working; the unmodified reference wrapper itself also runs on these
kernels through `pytorch3d_amd.shim`.  What differs is the host side: everything is derived from the packed layout, the
radius may also be given per packed point, and a scalar radius is materialised on the device.
"""
import math
from typing import Optional, Sequence, Tuple, Union

import torch

from . import _C
from .rasterize_meshes import parse_image_size

MAX_BINS_PER_SIDE = 22  # rasterize_points.py:21 (kMaxPointsPerBin): bins along the longer image side must stay below this

Radius = Union[float, int, Sequence[float], torch.Tensor]


def default_bin_size(longest_side: int) -> int:
    """The reference's choice when bin_size is None (rasterize_points.py:104-113): 2^max(ceil(log2(side)) - 4, 4)."""
    return 1 << max(int(math.ceil(math.log2(longest_side))) - 4, 4)


def radius_per_packed_point(radius: Radius, clouds) -> torch.Tensor:
    """One float32 radius per packed point, on the points' device.

    scalar                  every point
    tensor / list (P_pack,) already per packed point
    tensor / list (N, P)    the reference's padded form (rasterize_points.py:145-184); P = the largest cloud.  A 1-D
                            input of length P with a single cloud is read as (1, P), as the reference does.
    """
    pts = clouds.points_packed()
    n_packed = pts.shape[0]
    if isinstance(radius, (float, int)) and not isinstance(radius, bool):
        return torch.full((n_packed,), float(radius), dtype=pts.dtype, device=pts.device)
    if isinstance(radius, (list, tuple)):
        radius = torch.as_tensor(radius)
    if not torch.is_tensor(radius):
        raise ValueError("radius must be a float, list, tuple or tensor; got %s" % type(radius))
    r = radius.to(device=pts.device, dtype=pts.dtype)
    if r.dim() == 0:
        return r.expand(n_packed).contiguous()
    n_clouds, p_max = len(clouds), clouds._P
    if r.dim() == 1 and n_clouds == 1 and r.shape[0] == p_max:
        r = r[None]
    if r.dim() == 1 and r.shape[0] == n_packed:
        return r.contiguous()
    if tuple(r.shape) != (n_clouds, p_max):
        raise ValueError("radius must be of shape (N, P): got %s" % repr(tuple(radius.shape)))
    return r.reshape(-1)[clouds.padded_to_packed_idx()].contiguous()


def rasterize_points(pointclouds, image_size: Union[int, Sequence[int]] = 256, radius: Radius = 0.01,
                     points_per_pixel: int = 8, bin_size: Optional[int] = None,
                     max_points_per_bin: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
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
    return _PointFragments.apply(pointclouds.points_packed(), radius_per_packed_point(radius, pointclouds),
                                 pointclouds.cloud_to_packed_first_idx(), pointclouds.num_points_per_cloud(),
                                 (size, int(points_per_pixel), int(bin_size), int(max_points_per_bin)))


class _PointFragments(torch.autograd.Function):
    """(points, radius, first, count, static) -> (idx, zbuf, dists2); gradient to the points only (the radius has none in
    the reference either, rasterize_points.py:236-242)."""

    @staticmethod
    def forward(ctx, points, radius, first, count, static):
        size, k, bin_size, cap = static
        idx, zbuf, dists2 = _C.rasterize_points(points, first, count, size, radius, k, bin_size, cap)
        ctx.save_for_backward(points, idx)
        ctx.mark_non_differentiable(idx)
        return idx, zbuf, dists2

    @staticmethod
    def backward(ctx, _g_idx, g_zbuf, g_dists2):
        points, idx = ctx.saved_tensors
        return _C.rasterize_points_backward(points, idx, g_zbuf, g_dists2), None, None, None, None
