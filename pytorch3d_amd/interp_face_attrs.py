"""`interpolate_face_attributes` on the HIP kernels (same call as pytorch3d/ops/interp_face_attrs.py:15-57).

out[n, y, x, k, :] = sum_i bary[n, y, x, k, i] * face_attributes[pix_to_face[n, y, x, k], i, :], zero where pix_to_face < 0.
The autograd node remembers the (N, H, W, K) image shape, which the reference flattens away: the backward maps lanes to
8x8 pixel tiles (p3d_interp_face_attrs_backward_nhwk) so that neighbouring samples of one face merge before they reach
memory.
"""
import torch

from . import _C


class _Interpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat_faces, flat_bary, attrs, shape):
        ctx.shape = shape
        ctx.save_for_backward(flat_faces, flat_bary, attrs)
        return _C.interp_face_attrs_forward(flat_faces, flat_bary, attrs)

    @staticmethod
    def backward(ctx, g_out):
        flat_faces, flat_bary, attrs = ctx.saved_tensors
        g_bary, g_attrs = _C.interp_face_attrs_backward(flat_faces, flat_bary, attrs, g_out.contiguous(), image_shape=ctx.shape)
        return None, g_bary, g_attrs, None


def interpolate_face_attributes(pix_to_face: torch.Tensor, barycentric_coords: torch.Tensor,
                                face_attributes: torch.Tensor) -> torch.Tensor:
    """pix_to_face (N,H,W,K) int64, barycentric_coords (N,H,W,K,3), face_attributes (F,3,D) -> (N,H,W,K,D)."""
    if face_attributes.dim() != 3 or face_attributes.shape[1] != 3:
        raise ValueError("Faces can only have three vertices; got %r" % (face_attributes.shape[1] if face_attributes.dim() > 1 else None))
    shape = tuple(barycentric_coords.shape[:4])
    if tuple(pix_to_face.shape) != shape:
        raise ValueError("pix_to_face must have shape (batch_size, H, W, K); got %r" % (tuple(pix_to_face.shape),))
    n_samples = shape[0] * shape[1] * shape[2] * shape[3]
    out = _Interpolate.apply(pix_to_face.reshape(n_samples), barycentric_coords.reshape(n_samples, 3), face_attributes, shape)
    return out.view(*shape, face_attributes.shape[2])
