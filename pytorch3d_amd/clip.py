"""Host-side mirror of pytorch3d/renderer/mesh/clip.py over the C ABI.

`ClipFrustum`, `ClippedFaces`, `clip_faces` and `convert_clipped_rasterization_to_original_faces` have the
reference's names, fields, return conventions, early exits and autograd behaviour; the ~40 torch kernels and
2-3 host syncs of the reference's clip_faces are three launches and ONE sync here (the output sizes depend on
the data, so one read-back is unavoidable), and the masked gather / bmm / scatter of the conversion is one
kernel forward and one backward (include/p3d_amd.h: p3d_clip_faces_*, p3d_convert_clipped_*).
"""
import ctypes
from typing import Optional, Tuple

import torch

from . import _C, _lib


class ClippedFaces:
    """clip.py:35-94 (same fields)."""

    __slots__ = ["face_verts", "mesh_to_face_first_idx", "num_faces_per_mesh", "faces_clipped_to_unclipped_idx",
                 "barycentric_conversion", "faces_clipped_to_conversion_idx", "clipped_faces_neighbor_idx"]

    def __init__(self, face_verts, mesh_to_face_first_idx, num_faces_per_mesh, faces_clipped_to_unclipped_idx=None,
                 barycentric_conversion=None, faces_clipped_to_conversion_idx=None, clipped_faces_neighbor_idx=None):
        self.face_verts = face_verts
        self.mesh_to_face_first_idx = mesh_to_face_first_idx
        self.num_faces_per_mesh = num_faces_per_mesh
        self.faces_clipped_to_unclipped_idx = faces_clipped_to_unclipped_idx
        self.barycentric_conversion = barycentric_conversion
        self.faces_clipped_to_conversion_idx = faces_clipped_to_conversion_idx
        self.clipped_faces_neighbor_idx = clipped_faces_neighbor_idx


class ClipFrustum:
    """clip.py:97-154 (same fields and defaults)."""

    __slots__ = ["left", "right", "top", "bottom", "znear", "zfar", "perspective_correct", "cull", "z_clip_value"]

    def __init__(self, left: Optional[float] = None, right: Optional[float] = None, top: Optional[float] = None,
                 bottom: Optional[float] = None, znear: Optional[float] = None, zfar: Optional[float] = None,
                 perspective_correct: bool = False, cull: bool = True, z_clip_value: Optional[float] = None) -> None:
        self.left = left
        self.right = right
        self.top = top
        self.bottom = bottom
        self.znear = znear
        self.zfar = zfar
        self.perspective_correct = perspective_correct
        self.cull = cull
        self.z_clip_value = z_clip_value


class _ClipEmit(torch.autograd.Function):
    """(face_verts) -> (face_verts_clipped, barycentric_conversion); the index tables ride along as non-differentiable
    outputs so that everything is produced by one launch."""

    @staticmethod
    def forward(ctx, face_verts, mesh_first, plan, totals, z_clip, persp):
        Fc, T3, T4, F = totals
        N = mesh_first.shape[0]
        dev = face_verts.device
        T = T3 + 2 * T4
        lib = _lib.load()
        with torch.cuda.device(dev):
            out_fv = torch.empty((Fc, 3, 3), dtype=torch.float32, device=dev)
            first_c = torch.empty((N,), dtype=torch.int64, device=dev)
            count_c = torch.empty((N,), dtype=torch.int64, device=dev)
            c2u = torch.empty((Fc,), dtype=torch.int64, device=dev)
            conv = torch.empty((T, 3, 3), dtype=torch.float32, device=dev)
            conv_idx = torch.empty((Fc if T else 0,), dtype=torch.int64, device=dev)
            nbr = torch.empty((Fc if T else 0,), dtype=torch.int64, device=dev)
            rc = lib.p3d_clip_faces_emit(_C._ptr(face_verts), F, _C._ptr(mesh_first), N, _C._ptr(plan), plan.numel(), Fc,
                                         T3, T4, float(z_clip), int(bool(persp)), _C._ptr(out_fv), _C._ptr(first_c),
                                         _C._ptr(count_c), _C._ptr(c2u), _C._ptr(conv), _C._ptr(conv_idx), _C._ptr(nbr),
                                         _C._stream(dev))
            _lib.check(rc, "clip_faces")
        ctx.save_for_backward(face_verts, plan)
        ctx.meta = (F, T3, T4, float(z_clip), int(bool(persp)))
        ctx.out_shape = (Fc, 3, 3)
        ctx.mark_non_differentiable(first_c, count_c, c2u, conv_idx, nbr)
        ctx.set_materialize_grads(False)
        return out_fv, conv, first_c, count_c, c2u, conv_idx, nbr

    @staticmethod
    def backward(ctx, g_fv, g_conv, *unused):
        face_verts, plan = ctx.saved_tensors
        F, T3, T4, z_clip, persp = ctx.meta
        dev = face_verts.device
        if g_fv is None and g_conv is None:
            return (None,) * 6
        lib = _lib.load()
        with torch.cuda.device(dev):
            if g_fv is None:  # only the conversion matrices fed the loss
                g_fv = torch.zeros(ctx.out_shape, dtype=torch.float32, device=dev)
            g_fv = g_fv.contiguous()
            g_conv_c = g_conv.contiguous() if g_conv is not None and g_conv.numel() else None
            out = torch.empty((F, 3, 3), dtype=torch.float32, device=dev)
            rc = lib.p3d_clip_faces_backward(_C._ptr(face_verts), F, _C._ptr(plan), plan.numel(), T3, T4, z_clip, persp,
                                             _C._ptr(g_fv), _C._ptr(g_conv_c), _C._ptr(out), _C._stream(dev))
            _lib.check(rc, "clip_faces_backward")
        return out, None, None, None, None, None


def clip_faces(face_verts_unclipped: torch.Tensor, mesh_to_face_first_idx: torch.Tensor,
               num_faces_per_mesh: torch.Tensor, frustum: ClipFrustum) -> ClippedFaces:
    """clip.py:324-615: cull faces outside the frustum, clip faces that cross z = z_clip_value."""
    dev = _C._same_device(("face_verts", face_verts_unclipped), ("mesh_to_face_first_idx", mesh_to_face_first_idx),
                          ("num_faces_per_mesh", num_faces_per_mesh))
    _C._check_face_verts(face_verts_unclipped)
    fv = _C._c(face_verts_unclipped, torch.float32)
    first = _C._c(mesh_to_face_first_idx, torch.int64)
    F = fv.shape[0]
    vals = [frustum.left, frustum.right, frustum.top, frustum.bottom, frustum.znear, frustum.zfar]
    mask = sum(1 << i for i, v in enumerate(vals) if v is not None)
    planes = (ctypes.c_float * 6)(*[0.0 if v is None else float(v) for v in vals])
    has_z = frustum.z_clip_value is not None
    z_clip = float(frustum.z_clip_value) if has_z else 0.0
    lib = _lib.load()
    with torch.cuda.device(dev):
        plan = torch.empty((int(lib.p3d_clip_faces_plan_bytes(F)),), dtype=torch.uint8, device=dev)
        rc = lib.p3d_clip_faces_plan(_C._ptr(fv), F, planes, mask, int(bool(frustum.cull)), int(has_z), z_clip,
                                     _C._ptr(plan), plan.numel(), _C._stream(dev))
        _lib.check(rc, "clip_faces")
        Fc, T3, T4, _ = (int(x) for x in plan[:32].view(torch.int64).tolist())  # the one host sync
    if Fc == F and T3 == 0 and T4 == 0:
        # nothing culled, nothing clipped (clip.py:381-388)
        return ClippedFaces(face_verts=face_verts_unclipped, mesh_to_face_first_idx=mesh_to_face_first_idx,
                            num_faces_per_mesh=num_faces_per_mesh)
    out_fv, conv, first_c, count_c, c2u, conv_idx, nbr = _ClipEmit.apply(fv, first, plan, (Fc, T3, T4, F), z_clip,
                                                                         frustum.perspective_correct)
    if T3 + T4 == 0:
        # faces were culled but none clipped (clip.py:461-468)
        return ClippedFaces(face_verts=out_fv, mesh_to_face_first_idx=first_c, num_faces_per_mesh=count_c,
                            faces_clipped_to_unclipped_idx=c2u)
    return ClippedFaces(face_verts=out_fv, mesh_to_face_first_idx=first_c, num_faces_per_mesh=count_c,
                        faces_clipped_to_unclipped_idx=c2u, barycentric_conversion=conv,
                        faces_clipped_to_conversion_idx=conv_idx, clipped_faces_neighbor_idx=nbr)


class _ConvertClipped(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bary_clipped, conv, pix_to_face_clipped, c2u, conv_idx):
        dev = bary_clipped.device
        p2f = pix_to_face_clipped.contiguous()
        bary = bary_clipped.contiguous()
        S = p2f.numel()
        has_conv = conv is not None and conv.numel() > 0
        lib = _lib.load()
        with torch.cuda.device(dev):
            p2f_u = torch.empty_like(p2f)
            bary_u = torch.empty_like(bary)
            rc = lib.p3d_convert_clipped_forward(_C._ptr(p2f), _C._ptr(bary), _C._ptr(c2u),
                                                 _C._ptr(conv if has_conv else None),
                                                 _C._ptr(conv_idx if has_conv else None), S, _C._ptr(p2f_u),
                                                 _C._ptr(bary_u), _C._stream(dev))
            _lib.check(rc, "convert_clipped_rasterization_to_original_faces")
        ctx.save_for_backward(p2f, bary, conv if has_conv else torch.empty(0, device=dev),
                              conv_idx if has_conv else torch.empty(0, dtype=torch.int64, device=dev))
        ctx.has_conv = has_conv
        ctx.mark_non_differentiable(p2f_u)
        ctx.set_materialize_grads(False)
        return bary_u, p2f_u

    @staticmethod
    def backward(ctx, g_bary_u, _g_p2f):
        if g_bary_u is None:
            return None, None, None, None, None
        p2f, bary, conv, conv_idx = ctx.saved_tensors
        dev = bary.device
        S = p2f.numel()
        g = g_bary_u.contiguous()
        lib = _lib.load()
        with torch.cuda.device(dev):
            g_bary = torch.empty_like(bary)
            T = conv.shape[0] if ctx.has_conv else 0
            g_conv = torch.empty((T, 3, 3), dtype=torch.float32, device=dev) if ctx.has_conv else None
            rc = lib.p3d_convert_clipped_backward(_C._ptr(p2f), _C._ptr(bary), _C._ptr(conv if ctx.has_conv else None),
                                                  _C._ptr(conv_idx if ctx.has_conv else None), _C._ptr(g), S, T,
                                                  _C._ptr(g_bary), _C._ptr(g_conv), _C._stream(dev))
            _lib.check(rc, "convert_clipped_rasterization_to_original_faces (backward)")
        return g_bary, g_conv, None, None, None


def convert_clipped_rasterization_to_original_faces(pix_to_face_clipped, bary_coords_clipped,
                                                    clipped_faces: ClippedFaces) -> Tuple[torch.Tensor, torch.Tensor]:
    """clip.py:618-734: express pix_to_face / barycentrics of the clipped faces in terms of the original faces."""
    c2u = clipped_faces.faces_clipped_to_unclipped_idx
    if c2u is None or c2u.numel() == 0:
        return pix_to_face_clipped, bary_coords_clipped
    bary_u, p2f_u = _ConvertClipped.apply(bary_coords_clipped, clipped_faces.barycentric_conversion, pix_to_face_clipped,
                                          c2u, clipped_faces.faces_clipped_to_conversion_idx)
    return p2f_u, bary_u
