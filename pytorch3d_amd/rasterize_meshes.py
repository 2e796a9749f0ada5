"""Host-side mirror of pytorch3d/renderer/mesh/rasterize_meshes.py:32-357 over pytorch3d_amd._C.

Same function name, arguments, defaults, heuristics and error messages as the reference's L2
entry point, so the parity tests read like the reference's own tests.  Frustum culling / z-plane
clipping (`z_clip_value` / `cull_to_frustum`; pure torch in the reference, pytorch3d/renderer/mesh/clip.py)
runs on the HIP kernels of pytorch3d_amd.clip.
"""
from typing import List, Optional, Tuple, Union

import os

import numpy as np
import torch

from . import _C

# the fused gather also writes the backward's per-face reciprocals (P3D_FACE_PRE=0: the backward forms them per sample)
FACE_PRE = os.environ.get("P3D_FACE_PRE", "1") not in ("", "0")

kMaxFacesPerBin = 22  # rasterize_meshes.py:29


def parse_image_size(image_size):
    """pytorch3d/renderer/utils.py parse_image_size: int or (H, W) -> (H, W)."""
    if not isinstance(image_size, (tuple, list)):
        return (int(image_size), int(image_size))
    if len(image_size) != 2:
        raise ValueError("Image size can only be a tuple/list of (H, W)")
    if not all(i > 0 for i in image_size):
        raise ValueError("Image sizes must be greater than 0; got %d, %d" % tuple(image_size))
    if not all(isinstance(i, int) for i in image_size):
        raise ValueError("Image sizes must be integers; got %r, %r" % tuple(image_size))
    return tuple(image_size)


def default_bin_size(max_image_size: int) -> int:
    """rasterize_meshes.py:195-210 (GPU branch)."""
    if max_image_size <= 64:
        return 8
    return int(2 ** max(np.ceil(np.log2(max_image_size)) - 4, 4))


def rasterize_meshes(
    meshes,
    image_size: Union[int, List[int], Tuple[int, int]] = 256,
    blur_radius: float = 0.0,
    faces_per_pixel: int = 8,
    bin_size: Optional[int] = None,
    max_faces_per_bin: Optional[int] = None,
    perspective_correct: bool = False,
    clip_barycentric_coords: bool = False,
    cull_backfaces: bool = False,
    z_clip_value: Optional[float] = None,
    cull_to_frustum: bool = False,
):
    """Returns (pix_to_face, zbuf, barycentric_coords, dists), each (N, H, W, faces_per_pixel[, 3])."""
    verts_packed = meshes.verts_packed()
    faces_packed = meshes.faces_packed()
    # without clipping the face gather, the rasterization and both their backwards run as one autograd node (the
    # per-face gradient is flushed straight to the vertices: no (F,3,3) intermediate, no separate scatter)
    fused_verts = (z_clip_value is None and not cull_to_frustum and _gatherable(verts_packed, faces_packed))
    face_verts = None if fused_verts else gather_face_verts(verts_packed, faces_packed)
    mesh_to_face_first_idx = meshes.mesh_to_faces_packed_first_idx()
    num_faces_per_mesh = meshes.num_faces_per_mesh()
    im_size = parse_image_size(image_size)
    max_image_size = max(*im_size)

    clipped_faces = None
    clipped_faces_neighbor_idx = None
    if z_clip_value is not None or cull_to_frustum:
        # rasterize_meshes.py:160-183: cull faces outside the view frustum, clip faces partially behind the camera
        from .clip import ClipFrustum, clip_faces

        frustum = ClipFrustum(left=-1, right=1, top=-1, bottom=1, perspective_correct=perspective_correct,
                              z_clip_value=z_clip_value, cull=cull_to_frustum)
        clipped_faces = clip_faces(face_verts, mesh_to_face_first_idx, num_faces_per_mesh, frustum=frustum)
        face_verts = clipped_faces.face_verts
        mesh_to_face_first_idx = clipped_faces.mesh_to_face_first_idx
        num_faces_per_mesh = clipped_faces.num_faces_per_mesh
        clipped_faces_neighbor_idx = clipped_faces.clipped_faces_neighbor_idx
    if clipped_faces_neighbor_idx is None:
        clipped_faces_neighbor_idx = torch.full(
            size=(faces_packed.shape[0] if fused_verts else face_verts.shape[0],), fill_value=-1,
            device=verts_packed.device, dtype=torch.int64)

    if bin_size is None:
        bin_size = default_bin_size(max_image_size)
    if bin_size != 0:
        faces_per_bin = 1 + (max_image_size - 1) // bin_size
        if faces_per_bin >= kMaxFacesPerBin:
            raise ValueError("bin_size too small, number of faces per bin must be less than %d; got %d" %
                             (kMaxFacesPerBin, faces_per_bin))
    if max_faces_per_bin is None:
        max_faces_per_bin = int(max(10000, meshes._F / 5))

    if fused_verts:
        return _RasterizeMeshVerts.apply(
            verts_packed, faces_packed, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx, im_size,
            blur_radius, faces_per_pixel, bin_size, max_faces_per_bin, perspective_correct, clip_barycentric_coords,
            cull_backfaces)
    pix_to_face, zbuf, barycentric_coords, dists = _RasterizeFaceVerts.apply(
        face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx, im_size, blur_radius,
        faces_per_pixel, bin_size, max_faces_per_bin, perspective_correct, clip_barycentric_coords, cull_backfaces)
    if clipped_faces is not None:
        # rasterize_meshes.py:239-249: fragments of the clipped faces -> fragments of the original faces
        from .clip import convert_clipped_rasterization_to_original_faces

        pix_to_face, barycentric_coords = convert_clipped_rasterization_to_original_faces(
            pix_to_face, barycentric_coords, clipped_faces)
    return pix_to_face, zbuf, barycentric_coords, dists


def _gatherable(verts_packed, faces_packed):
    return (verts_packed.is_cuda and verts_packed.dtype == torch.float32 and faces_packed.dtype == torch.int64
            and faces_packed.device == verts_packed.device and verts_packed.dim() == 2 and faces_packed.dim() == 2
            # the kernels move xyz triples of triangle corners: anything else ((V, C) attributes, quads) takes torch indexing
            and verts_packed.shape[1] == 3 and faces_packed.shape[1] == 3)


def gather_face_verts(verts_packed, faces_packed):
    """`verts_packed[faces_packed]` (rasterize_meshes.py:146) -> (F, 3, 3).  On the GPU both the gather and its
    autograd scatter are single HIP kernels (include/p3d_amd.h: p3d_gather_face_verts / p3d_scatter_face_grads)
    instead of torch indexing, whose backward on ROCm is a radix sort + segmented sum."""
    if _gatherable(verts_packed, faces_packed):
        return _GatherFaceVerts.apply(verts_packed, faces_packed)
    return verts_packed[faces_packed]


class _GatherFaceVerts(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, faces):
        import ctypes

        from . import _lib

        lib = _lib.load()
        verts_c, faces_c = verts.contiguous(), faces.contiguous()
        V, F = verts_c.shape[0], faces_c.shape[0]
        with torch.cuda.device(verts.device):
            out = torch.empty((F, 3, 3), dtype=torch.float32, device=verts.device)
            if F:
                rc = lib.p3d_gather_face_verts(_C._ptr(verts_c), _C._ptr(faces_c), V, F, _C._ptr(out),
                                               _C._stream(verts.device))
                _lib.check(rc, "gather_face_verts")
        ctx.save_for_backward(faces_c)
        ctx.V = V
        return out

    @staticmethod
    def backward(ctx, grad_face_verts):
        from . import _lib

        (faces,) = ctx.saved_tensors
        lib = _lib.load()
        g = grad_face_verts.contiguous()
        V, F = ctx.V, faces.shape[0]
        with torch.cuda.device(g.device):
            out = torch.empty((V, 3), dtype=torch.float32, device=g.device)
            if V:
                rc = lib.p3d_scatter_face_grads(_C._ptr(g), _C._ptr(faces), V, F, _C._ptr(out), _C._stream(g.device))
                _lib.check(rc, "scatter_face_grads")
        return out, None


class _RasterizeFaceVerts(torch.autograd.Function):
    """Autograd wrapper, as rasterize_meshes.py:252-357."""

    @staticmethod
    def forward(ctx, face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx,
                image_size=(256, 256), blur_radius=0.01, faces_per_pixel=0, bin_size=0, max_faces_per_bin=0,
                perspective_correct=False, clip_barycentric_coords=False, cull_backfaces=False):
        (pix_to_face, zbuf, barycentric_coords, dists), cover = _C._rasterize_meshes_covered(
            face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx, image_size,
            blur_radius, faces_per_pixel, bin_size, max_faces_per_bin, perspective_correct, clip_barycentric_coords,
            cull_backfaces)
        # the forward's row cover rides with pix_to_face (include/p3d_amd.h): the backward skips what the forward left empty
        ctx.save_for_backward(face_verts, pix_to_face, cover)
        ctx.mark_non_differentiable(pix_to_face)
        # do not let autograd materialise a zero "gradient" for the int64 pix_to_face (1 GB at the bench size)
        ctx.set_materialize_grads(False)
        ctx.perspective_correct = perspective_correct
        ctx.clip_barycentric_coords = clip_barycentric_coords
        return pix_to_face, zbuf, barycentric_coords, dists

    @staticmethod
    def backward(ctx, grad_pix_to_face, grad_zbuf, grad_barycentric_coords, grad_dists):
        face_verts, pix_to_face, cover = ctx.saved_tensors
        cover = _C.checked_cover(pix_to_face, cover)  # (P3D_CHECK=1: verified on the device before it is trusted)
        if grad_zbuf is None and grad_barycentric_coords is None and grad_dists is None:
            return (None,) * 12
        if grad_zbuf is None:
            grad_zbuf = torch.zeros(pix_to_face.shape, dtype=torch.float32, device=pix_to_face.device)
        if grad_dists is None:
            grad_dists = torch.zeros(pix_to_face.shape, dtype=torch.float32, device=pix_to_face.device)
        if grad_barycentric_coords is None:
            grad_barycentric_coords = torch.zeros(pix_to_face.shape + (3,), dtype=torch.float32,
                                                  device=pix_to_face.device)
        grad_face_verts = _C.rasterize_meshes_backward(face_verts, pix_to_face, grad_zbuf, grad_barycentric_coords,
                                                       grad_dists, ctx.perspective_correct,
                                                       ctx.clip_barycentric_coords, _cover=cover)
        return (grad_face_verts,) + (None,) * 11


class _RasterizeMeshVerts(torch.autograd.Function):
    """`verts_packed[faces_packed]` + _RasterizeFaceVerts as ONE node: forward = the gather kernel + the rasterizer,
    backward = p3d_rasterize_meshes_backward_verts, which flushes the per-face partials straight to grad_verts."""

    @staticmethod
    def forward(ctx, verts, faces, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx, image_size,
                blur_radius, faces_per_pixel, bin_size, max_faces_per_bin, perspective_correct, clip_barycentric_coords,
                cull_backfaces):
        from . import _lib

        lib = _lib.load()
        verts_c, faces_c = verts.contiguous(), faces.contiguous()
        V, F = verts_c.shape[0], faces_c.shape[0]
        with torch.cuda.device(verts.device):
            face_verts = torch.empty((F, 3, 3), dtype=torch.float32, device=verts.device)
            # per-face reciprocals for the backward (include/p3d_amd.h: p3d_gather_face_verts_pre), written by the gather
            face_pre = torch.empty((F, 4), dtype=torch.float32, device=verts.device) if (FACE_PRE and ctx.needs_input_grad[0]) else None
            if F and face_pre is not None:
                rc = lib.p3d_gather_face_verts_pre(_C._ptr(verts_c), _C._ptr(faces_c), V, F, _C._ptr(face_verts), _C._ptr(face_pre),
                                                   _C._stream(verts.device))
                _lib.check(rc, "gather_face_verts_pre")
            elif F:
                rc = lib.p3d_gather_face_verts(_C._ptr(verts_c), _C._ptr(faces_c), V, F, _C._ptr(face_verts),
                                               _C._stream(verts.device))
                _lib.check(rc, "gather_face_verts")
        (pix_to_face, zbuf, barycentric_coords, dists), cover = _C._rasterize_meshes_covered(
            face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx, image_size, blur_radius,
            faces_per_pixel, bin_size, max_faces_per_bin, perspective_correct, clip_barycentric_coords, cull_backfaces)
        ctx.save_for_backward(face_verts, faces_c, pix_to_face, cover, face_pre)
        ctx.mark_non_differentiable(pix_to_face)
        ctx.set_materialize_grads(False)
        ctx.V = V
        ctx.flags = (int(bool(perspective_correct)), int(bool(clip_barycentric_coords)))
        return pix_to_face, zbuf, barycentric_coords, dists

    @staticmethod
    def backward(ctx, grad_pix_to_face, grad_zbuf, grad_barycentric_coords, grad_dists):
        from . import _lib

        face_verts, faces, pix_to_face, cover, face_pre = ctx.saved_tensors

        cover = _C.checked_cover(pix_to_face, cover)  # (P3D_CHECK=1: verified on the device before it is trusted)
        if grad_zbuf is None and grad_barycentric_coords is None and grad_dists is None:
            return (None,) * 13
        _refuse_when_deterministic()
        dev = pix_to_face.device
        N, H, W, K = pix_to_face.shape
        zeros = lambda *tail: torch.zeros(tuple(pix_to_face.shape) + tail, dtype=torch.float32, device=dev)
        gz = grad_zbuf.contiguous() if grad_zbuf is not None else zeros()
        gd = grad_dists.contiguous() if grad_dists is not None else zeros()
        gb = grad_barycentric_coords.contiguous() if grad_barycentric_coords is not None else zeros(3)
        lib = _lib.load()
        with torch.cuda.device(dev):
            grad_verts = torch.empty((ctx.V, 3), dtype=torch.float32, device=dev)
            has_list = _C.cover_has_list(cover, N, H, W)
            if face_pre is not None and (has_list or cover is None):
                rc = lib.p3d_rasterize_meshes_backward_verts_pre(
                    _C._ptr(face_verts), _C._ptr(face_pre), _C._ptr(faces), _C._ptr(pix_to_face), _C._ptr(gz), _C._ptr(gb), _C._ptr(gd),
                    _C.cover_ptr(cover, N, H, W), faces.shape[0], ctx.V, N, H, W, K, ctx.flags[0], ctx.flags[1], _C._ptr(grad_verts),
                    _C._stream(dev))
            elif has_list:  # the forward listed the areas that hold a face: no list builder, no workspace
                rc = lib.p3d_rasterize_meshes_backward_verts_with_cover_list(
                    _C._ptr(face_verts), _C._ptr(faces), _C._ptr(pix_to_face), _C._ptr(gz), _C._ptr(gb), _C._ptr(gd),
                    _C.cover_ptr(cover, N, H, W), faces.shape[0], ctx.V, N, H, W, K, ctx.flags[0], ctx.flags[1], _C._ptr(grad_verts),
                    _C._stream(dev))
            else:
                ws = _C.backward_workspace(cover, N, H, W, dev)
                rc = lib.p3d_rasterize_meshes_backward_verts_with_cover(
                    _C._ptr(face_verts), _C._ptr(faces), _C._ptr(pix_to_face), _C._ptr(gz), _C._ptr(gb), _C._ptr(gd),
                    _C.cover_ptr(cover, N, H, W), faces.shape[0], ctx.V, N, H, W, K, ctx.flags[0], ctx.flags[1], _C._ptr(grad_verts),
                    _C._ptr(ws), ws.numel(), _C._stream(dev))
            _lib.check(rc, "rasterize_meshes_backward")
        return (grad_verts,) + (None,) * 12


# ---------------------------------------------------------------------------------------------------------------------
# World-space entry: the camera transform fused into the face gather
# ---------------------------------------------------------------------------------------------------------------------
def _pack_matrices(world_to_view, view_to_ndc, n_meshes, device):
    """(N or 1, 4, 4) x 2 -> (num, 2, 4, 4) float32 contiguous, num = N or 1.  Row-vector convention, as
    Transform3d.get_matrix() returns them (pytorch3d/transforms/transform3d.py)."""
    mats = []
    for m in (world_to_view, view_to_ndc):
        m = torch.as_tensor(m, dtype=torch.float32, device=device)
        if m.dim() == 2:
            m = m[None]
        if m.dim() != 3 or m.shape[1:] != (4, 4) or m.shape[0] not in (1, n_meshes):
            raise ValueError("camera matrices must have shape (4, 4), (1, 4, 4) or (N, 4, 4); got %s" % repr(tuple(m.shape)))
        mats.append(m)
    num = max(mats[0].shape[0], mats[1].shape[0])
    return torch.stack([m.expand(num, 4, 4) for m in mats], 1).contiguous()


def transform_points_reference(verts_packed, mesh_idx, world_to_view, view_to_ndc):
    """MeshRasterizer.transform (renderer/mesh/rasterizer.py:171-216) with torch ops on packed vertices: the differentiable
    path (cameras that require grad) and the oracle the HIP kernels are tested against."""
    one = torch.ones_like(verts_packed[:, :1])
    a = world_to_view[mesh_idx] if world_to_view.shape[0] > 1 else world_to_view.expand(verts_packed.shape[0], 4, 4)
    b = view_to_ndc[mesh_idx] if view_to_ndc.shape[0] > 1 else view_to_ndc.expand(verts_packed.shape[0], 4, 4)
    vh = torch.bmm(torch.cat([verts_packed, one], 1)[:, None], a)[:, 0]
    view = vh[:, :3] / vh[:, 3:]
    nh = torch.bmm(torch.cat([view, one], 1)[:, None], b)[:, 0]
    ndc = nh[:, :3] / nh[:, 3:]
    return torch.cat([ndc[:, :2], view[:, 2:3]], 1)


def rasterize_meshes_world(meshes_world, world_to_view, view_to_ndc, image_size=256, blur_radius: float = 0.0,
                           faces_per_pixel: int = 8, bin_size: Optional[int] = None, max_faces_per_bin: Optional[int] = None,
                           perspective_correct: bool = False, clip_barycentric_coords: bool = False,
                           cull_backfaces: bool = False):
    """`MeshRasterizer.transform` + `rasterize_meshes` on world-space vertices (no z-clipping / frustum culling).

    world_to_view, view_to_ndc: (N, 4, 4) or (1, 4, 4) matrices, `cameras.get_world_to_view_transform().get_matrix()` and
    `cameras.get_projection_transform().compose(cameras.get_ndc_camera_transform()).get_matrix()`.  The transform runs
    inside the face gather (one launch); the backward delivers the gradient wrt the WORLD vertices.  Matrices that
    require grad take the torch formulation of the transform (gradients to the cameras through autograd), followed by
    the fused gather + rasterization."""
    verts = meshes_world.verts_packed()
    faces = meshes_world.faces_packed()
    n = len(meshes_world)
    w2v = torch.as_tensor(world_to_view)
    v2n = torch.as_tensor(view_to_ndc)
    cams_need_grad = (torch.is_tensor(world_to_view) and world_to_view.requires_grad) or \
                     (torch.is_tensor(view_to_ndc) and view_to_ndc.requires_grad)
    if cams_need_grad or not _gatherable(verts, faces):
        a = w2v.to(verts.device, torch.float32)
        b = v2n.to(verts.device, torch.float32)
        a = a[None] if a.dim() == 2 else a
        b = b[None] if b.dim() == 2 else b
        ndc = transform_points_reference(verts, meshes_world.verts_packed_to_mesh_idx(), a, b)
        # any object with the packed accessors is accepted here (PackedMeshes, the reference's Meshes): do not ask it for
        # more than rasterize_meshes itself does (update_verts_packed exists on PackedMeshes only)
        return rasterize_meshes(_PackedVertsView(meshes_world, ndc), image_size, blur_radius, faces_per_pixel, bin_size,
                                max_faces_per_bin, perspective_correct, clip_barycentric_coords, cull_backfaces)
    mats = _pack_matrices(w2v, v2n, n, verts.device)
    im_size = parse_image_size(image_size)
    max_image_size = max(*im_size)
    if bin_size is None:
        bin_size = default_bin_size(max_image_size)
    if bin_size != 0:
        faces_per_bin = 1 + (max_image_size - 1) // bin_size
        if faces_per_bin >= kMaxFacesPerBin:
            raise ValueError("bin_size too small, number of faces per bin must be less than %d; got %d" %
                             (kMaxFacesPerBin, faces_per_bin))
    if max_faces_per_bin is None:
        max_faces_per_bin = int(max(10000, meshes_world._F / 5))
    nbr = torch.full((faces.shape[0],), -1, dtype=torch.int64, device=verts.device)
    return _RasterizeMeshWorld.apply(verts, faces, meshes_world.mesh_to_faces_packed_first_idx(),
                                     meshes_world.num_faces_per_mesh(), meshes_world.mesh_to_verts_packed_first_idx(), mats, nbr,
                                     (im_size, blur_radius, faces_per_pixel, bin_size, max_faces_per_bin,
                                      bool(perspective_correct), bool(clip_barycentric_coords), bool(cull_backfaces)))


class _TransformVerts(torch.autograd.Function):
    """world vertices (V,3) -> NDC x, y + view depth (V,3): MeshRasterizer.transform (rasterizer.py:171-216) on the PACKED
    vertices in one launch (p3d_transform_verts_forward), backward = p3d_transform_verts_backward."""

    @staticmethod
    def forward(ctx, verts, vert_first, mats):
        from . import _lib

        lib = _lib.load()
        v = verts.contiguous()
        dev = v.device
        with torch.cuda.device(dev):
            out = torch.empty_like(v)
            if v.shape[0]:
                rc = lib.p3d_transform_verts_forward(_C._ptr(v), _C._ptr(vert_first), _C._ptr(mats), v.shape[0], vert_first.shape[0],
                                                     mats.shape[0], _C._ptr(out), _C._stream(dev))
                _lib.check(rc, "transform_verts_forward")
        ctx.save_for_backward(v, vert_first, mats)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib

        v, vert_first, mats = ctx.saved_tensors
        lib = _lib.load()
        dev = v.device
        g = g.contiguous()
        with torch.cuda.device(dev):
            out = torch.empty_like(v)
            if v.shape[0]:
                rc = lib.p3d_transform_verts_backward(_C._ptr(v), _C._ptr(vert_first), _C._ptr(mats), _C._ptr(g), v.shape[0],
                                                      vert_first.shape[0], mats.shape[0], _C._ptr(out), _C._stream(dev))
                _lib.check(rc, "transform_verts_backward")
        return out, None, None


def transform_verts_to_ndc(meshes_world, world_to_view, view_to_ndc):
    """Packed world vertices of `meshes_world` -> packed NDC vertices (x, y in NDC, z = view depth), differentiable with
    respect to the vertices.  Matrices as in rasterize_meshes_world; they must not require grad (callers take the torch
    formulation then)."""
    verts = meshes_world.verts_packed()
    mats = _pack_matrices(torch.as_tensor(world_to_view), torch.as_tensor(view_to_ndc), len(meshes_world), verts.device)
    return _TransformVerts.apply(verts, meshes_world.mesh_to_verts_packed_first_idx().contiguous(), mats)


class _PackedVertsView:
    """The packed accessors `rasterize_meshes` reads, with the vertices replaced (topology shared with `meshes`)."""

    def __init__(self, meshes, verts_packed):
        self._meshes, self._verts = meshes, verts_packed

    def __len__(self):
        return len(self._meshes)

    def verts_packed(self):
        return self._verts

    def faces_packed(self):
        return self._meshes.faces_packed()

    def mesh_to_faces_packed_first_idx(self):
        return self._meshes.mesh_to_faces_packed_first_idx()

    def num_faces_per_mesh(self):
        return self._meshes.num_faces_per_mesh()

    @property
    def _F(self):
        return self._meshes._F


def _refuse_when_deterministic():
    """The backward scatters with float atomics, like the reference's (rasterize_meshes.cu:587 alertNotDeterministic)."""
    if torch.are_deterministic_algorithms_enabled() and not torch.is_deterministic_algorithms_warn_only_enabled():
        raise RuntimeError("RasterizeMeshesBackwardCuda does not have a deterministic implementation")


class _RasterizeMeshWorld(torch.autograd.Function):
    """world vertices -> fragments as ONE node: p3d_transform_gather_face_verts + the rasterizer; backward =
    p3d_rasterize_meshes_backward_verts (per-vertex NDC gradient) + p3d_transform_verts_backward."""

    @staticmethod
    def forward(ctx, verts, faces, face_first, num_faces, vert_first, mats, nbr, static):
        from . import _lib

        im_size, blur, K, bin_size, cap, persp, clip, cull = static
        lib = _lib.load()
        verts_c, faces_c = verts.contiguous(), faces.contiguous()
        V, F, N = verts_c.shape[0], faces_c.shape[0], face_first.shape[0]
        dev = verts.device
        with torch.cuda.device(dev):
            face_verts = torch.empty((F, 3, 3), dtype=torch.float32, device=dev)
            if F:
                rc = lib.p3d_transform_gather_face_verts(_C._ptr(verts_c), _C._ptr(faces_c), _C._ptr(face_first), _C._ptr(mats), V, F,
                                                         N, mats.shape[0], _C._ptr(face_verts), _C._stream(dev))
                _lib.check(rc, "transform_gather_face_verts")
        out, cover = _C._rasterize_meshes_covered(face_verts, face_first, num_faces, nbr, im_size, blur, K, bin_size, cap, persp, clip,
                                                  cull)
        ctx.save_for_backward(verts_c, faces_c, vert_first, mats, face_verts, out[0], cover)
        ctx.mark_non_differentiable(out[0])
        ctx.set_materialize_grads(False)
        ctx.flags = (int(persp), int(clip))
        return out

    @staticmethod
    def backward(ctx, _g_idx, grad_zbuf, grad_bary, grad_dists):
        from . import _lib

        verts, faces, vert_first, mats, face_verts, pix_to_face, cover = ctx.saved_tensors
        cover = _C.checked_cover(pix_to_face, cover)  # (P3D_CHECK=1: verified on the device before it is trusted)
        if grad_zbuf is None and grad_bary is None and grad_dists is None:
            return (None,) * 8
        _refuse_when_deterministic()
        dev = pix_to_face.device
        N, H, W, K = pix_to_face.shape
        zeros = lambda *tail: torch.zeros(tuple(pix_to_face.shape) + tail, dtype=torch.float32, device=dev)
        gz = grad_zbuf.contiguous() if grad_zbuf is not None else zeros()
        gd = grad_dists.contiguous() if grad_dists is not None else zeros()
        gb = grad_bary.contiguous() if grad_bary is not None else zeros(3)
        lib = _lib.load()
        V = verts.shape[0]
        with torch.cuda.device(dev):
            g_ndc = torch.empty((V, 3), dtype=torch.float32, device=dev)
            if _C.cover_has_list(cover, N, H, W):
                rc = lib.p3d_rasterize_meshes_backward_verts_with_cover_list(
                    _C._ptr(face_verts), _C._ptr(faces), _C._ptr(pix_to_face), _C._ptr(gz), _C._ptr(gb), _C._ptr(gd),
                    _C.cover_ptr(cover, N, H, W), faces.shape[0], V, N, H, W, K, ctx.flags[0], ctx.flags[1], _C._ptr(g_ndc), _C._stream(dev))
            else:
                ws = _C.backward_workspace(cover, N, H, W, dev)
                rc = lib.p3d_rasterize_meshes_backward_verts_with_cover(
                    _C._ptr(face_verts), _C._ptr(faces), _C._ptr(pix_to_face), _C._ptr(gz), _C._ptr(gb), _C._ptr(gd),
                    _C.cover_ptr(cover, N, H, W), faces.shape[0], V, N, H, W, K, ctx.flags[0], ctx.flags[1], _C._ptr(g_ndc), _C._ptr(ws),
                    ws.numel(), _C._stream(dev))
            _lib.check(rc, "rasterize_meshes_backward")
            g_world = torch.empty((V, 3), dtype=torch.float32, device=dev)
            rc = lib.p3d_transform_verts_backward(_C._ptr(verts), _C._ptr(vert_first), _C._ptr(mats), _C._ptr(g_ndc), V,
                                                  vert_first.shape[0], mats.shape[0], _C._ptr(g_world), _C._stream(dev))
            _lib.check(rc, "transform_verts_backward")
        return (g_world,) + (None,) * 7
