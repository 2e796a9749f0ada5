"""Host-side mirror of pytorch3d/ops/interp_face_attrs.py:15-83 over pytorch3d_amd._C."""
import torch

from . import _C


def interpolate_face_attributes(pix_to_face: torch.Tensor, barycentric_coords: torch.Tensor,
                                face_attributes: torch.Tensor) -> torch.Tensor:
    """pix_to_face (N,H,W,K) i64, barycentric_coords (N,H,W,K,3), face_attributes (F,3,D) -> (N,H,W,K,D)."""
    F, FV, D = face_attributes.shape
    if FV != 3:
        raise ValueError("Faces can only have three vertices; got %r" % FV)
    N, H, W, K, _ = barycentric_coords.shape
    if pix_to_face.shape != (N, H, W, K):
        raise ValueError("pix_to_face must have shape (batch_size, H, W, K); got %r" % (tuple(pix_to_face.shape),))
    pix_to_face = pix_to_face.reshape(-1)
    barycentric_coords = barycentric_coords.reshape(N * H * W * K, 3)
    out = _InterpFaceAttrs.apply(pix_to_face, barycentric_coords, face_attributes, (N, H, W, K))
    return out.view(N, H, W, K, -1)


class _InterpFaceAttrs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pix_to_face, barycentric_coords, face_attrs, image_shape=None):
        ctx.save_for_backward(pix_to_face, barycentric_coords, face_attrs)
        ctx.image_shape = image_shape  # lets the backward map lanes to pixel tiles (the reference flattens and forgets)
        return _C.interp_face_attrs_forward(pix_to_face, barycentric_coords, face_attrs)

    @staticmethod
    def backward(ctx, grad_pix_attrs):
        pix_to_face, barycentric_coords, face_attrs = ctx.saved_tensors
        grad_bary, grad_face_attrs = _C.interp_face_attrs_backward(pix_to_face, barycentric_coords, face_attrs,
                                                                    grad_pix_attrs.contiguous(),
                                                                    image_shape=ctx.image_shape)
        return None, grad_bary, grad_face_attrs, None
