"""Host-side mirror of the texture classes' sample_textures over the C ABI.

TexturesAtlas.sample_textures (pytorch3d/renderer/mesh/textures.py:565-612): `sample_textures_atlas`, one kernel each way
(include/p3d_amd.h: p3d_sample_atlas_forward / _backward), gradient to the atlas only -- as in the reference.

TexturesUV.sample_textures (textures.py:1190-1313): `sample_textures_uv`, uv interpolation + grid_sample fused into one
kernel each way -- with one map per mesh (p3d_sample_uv_forward / _backward) or, given `maps_ids`, several
(p3d_sample_uv_multi_forward / _backward: the reference's 3-D grid_sample over the mesh's maps).  Padding "zeros" or
"border", sampling "bilinear" or "nearest", as F.grid_sample defines them.  Gradients flow to the maps, the per-face uvs
and the barycentric coordinates.
"""
import torch

from . import _C, _lib

_PAD = {"zeros": 0, "border": 1}
_MODE = {"bilinear": 0, "nearest": 1}


class _SampleUV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pix_to_face, bary, face_uvs, maps, align, pad, mode):
        N, H, W, K = pix_to_face.shape
        dev = bary.device
        p2f, b, fu, mp = pix_to_face.contiguous(), bary.contiguous(), face_uvs.contiguous(), maps.contiguous()
        Nm, Hm, Wm, C = mp.shape
        F = fu.shape[0]
        lib = _lib.load()
        with torch.cuda.device(dev):
            out = torch.empty((N, H, W, K, C), dtype=torch.float32, device=dev)
            if out.numel():
                rc = lib.p3d_sample_uv_forward(_C._ptr(p2f), _C._ptr(b), _C._ptr(fu), _C._ptr(mp), N, H, W, K, F, Hm, Wm, C,
                                               align, pad, mode, _C._ptr(out), _C._stream(dev))
                _lib.check(rc, "sample_textures")
        ctx.save_for_backward(p2f, b, fu, mp)
        ctx.cfg = (align, pad, mode)
        return out

    @staticmethod
    def backward(ctx, grad_texels):
        p2f, b, fu, mp = ctx.saved_tensors
        align, pad, mode = ctx.cfg
        N, H, W, K = p2f.shape
        Nm, Hm, Wm, C = mp.shape
        F = fu.shape[0]
        dev = b.device
        g = grad_texels.contiguous()
        lib = _lib.load()
        with torch.cuda.device(dev):
            gb = torch.empty((N, H, W, K, 3), dtype=torch.float32, device=dev)
            gfu = torch.empty((F, 3, 2), dtype=torch.float32, device=dev)
            gm = torch.empty((Nm, Hm, Wm, C), dtype=torch.float32, device=dev)
            rc = lib.p3d_sample_uv_backward(_C._ptr(g), _C._ptr(p2f), _C._ptr(b), _C._ptr(fu), _C._ptr(mp), N, H, W, K, F, Hm,
                                            Wm, C, align, pad, mode, _C._ptr(gb), _C._ptr(gfu), _C._ptr(gm), _C._stream(dev))
            _lib.check(rc, "sample_textures_backward")
        return None, gb, gfu, gm, None, None, None


class _SampleUVMulti(torch.autograd.Function):
    """The maps_ids form: maps (N,M,Hm,Wm,C), ids = maps_ids_padded flattened (include/p3d_amd.h: p3d_sample_uv_multi_*)."""

    @staticmethod
    def forward(ctx, pix_to_face, bary, face_uvs, maps, ids, align, pad, mode):
        N, H, W, K = pix_to_face.shape
        dev = bary.device
        p2f, b, fu, mp, idv = pix_to_face.contiguous(), bary.contiguous(), face_uvs.contiguous(), maps.contiguous(), ids.contiguous()
        _, M, Hm, Wm, C = mp.shape
        F, L = fu.shape[0], idv.numel()
        lib = _lib.load()
        with torch.cuda.device(dev):
            out = torch.empty((N, H, W, K, C), dtype=torch.float32, device=dev)
            if out.numel():
                rc = lib.p3d_sample_uv_multi_forward(_C._ptr(p2f), _C._ptr(b), _C._ptr(fu), _C._ptr(mp), _C._ptr(idv), L, N, H, W,
                                                     K, F, M, Hm, Wm, C, align, pad, mode, _C._ptr(out), _C._stream(dev))
                _lib.check(rc, "sample_textures (maps_ids)")
        ctx.save_for_backward(p2f, b, fu, mp, idv)
        ctx.cfg = (align, pad, mode)
        return out

    @staticmethod
    def backward(ctx, grad_texels):
        p2f, b, fu, mp, idv = ctx.saved_tensors
        align, pad, mode = ctx.cfg
        N, H, W, K = p2f.shape
        _, M, Hm, Wm, C = mp.shape
        F, L = fu.shape[0], idv.numel()
        dev = b.device
        g = grad_texels.contiguous()
        lib = _lib.load()
        with torch.cuda.device(dev):
            gb = torch.empty((N, H, W, K, 3), dtype=torch.float32, device=dev)
            gfu = torch.empty((F, 3, 2), dtype=torch.float32, device=dev)
            gm = torch.empty_like(mp)
            rc = lib.p3d_sample_uv_multi_backward(_C._ptr(g), _C._ptr(p2f), _C._ptr(b), _C._ptr(fu), _C._ptr(mp), _C._ptr(idv), L,
                                                  N, H, W, K, F, M, Hm, Wm, C, align, pad, mode, _C._ptr(gb), _C._ptr(gfu),
                                                  _C._ptr(gm), _C._stream(dev))
            _lib.check(rc, "sample_textures_backward (maps_ids)")
        return None, gb, gfu, gm, None, None, None, None


def sample_textures_uv(fragments, faces_verts_uvs, maps, align_corners: bool = True, padding_mode: str = "border",
                       sampling_mode: str = "bilinear", maps_ids=None) -> torch.Tensor:
    """TexturesUV.sample_textures(fragments) for faces_verts_uvs = cat(verts_uvs_list[i][faces_uvs_list[i]]) (F,3,2) and
    maps = maps_padded() (N,Hm,Wm,C) -> texels (N,H,W,K,C).  Defaults as TexturesUV.__init__ (textures.py:716-718).

    maps_ids (several maps per mesh, textures.py:736-744): maps is (N,M,Hm,Wm,C) and maps_ids = maps_ids_padded()
    (N,Fmax) int64 -- flattened and indexed by the packed face index, as the reference does (textures.py:1283-1289: the
    two layouts agree when all meshes of the batch have Fmax faces).  The reference samples the maps as a 3-D volume:
    "bilinear" also blends neighbouring maps where the un-normalised map coordinate is not an integer."""
    if maps_ids is not None:
        return _sample_textures_uv_multi(fragments, faces_verts_uvs, maps, maps_ids, align_corners, padding_mode, sampling_mode)
    if padding_mode not in _PAD:
        raise NotImplementedError(f"sample_textures_uv: padding_mode {padding_mode!r} (only 'zeros' and 'border')")
    if sampling_mode not in _MODE:
        raise ValueError(f"sample_textures_uv: sampling_mode {sampling_mode!r}")
    _C._check_fragments("sample_textures_uv", fragments.pix_to_face, bary_coords=fragments.bary_coords,
                        faces_verts_uvs=faces_verts_uvs, maps=maps)
    if faces_verts_uvs.dim() != 3 or faces_verts_uvs.shape[1:] != (3, 2):
        raise ValueError("faces_verts_uvs must have shape (F, 3, 2)")
    if maps.dim() != 4 or maps.shape[0] != fragments.pix_to_face.shape[0]:
        raise ValueError("maps must have shape (N, H, W, C) with one map per batch element")
    if faces_verts_uvs.dtype != torch.float32 or maps.dtype != torch.float32:
        raise RuntimeError("sample_textures_uv: faces_verts_uvs and maps must be float32")
    return _SampleUV.apply(fragments.pix_to_face, fragments.bary_coords, faces_verts_uvs, maps, int(bool(align_corners)),
                           _PAD[padding_mode], _MODE[sampling_mode])


def _sample_textures_uv_multi(fragments, faces_verts_uvs, maps, maps_ids, align_corners, padding_mode, sampling_mode):
    if padding_mode not in _PAD:
        raise NotImplementedError(f"sample_textures_uv: padding_mode {padding_mode!r} (only 'zeros' and 'border')")
    if sampling_mode not in _MODE:
        raise ValueError(f"sample_textures_uv: sampling_mode {sampling_mode!r}")
    _C._check_fragments("sample_textures_uv", fragments.pix_to_face, bary_coords=fragments.bary_coords,
                        faces_verts_uvs=faces_verts_uvs, maps=maps)
    _C._need_gpu(maps_ids, "maps_ids")
    if maps_ids.device != fragments.pix_to_face.device:
        raise RuntimeError(f"Expected all tensors to be on the same GPU, but maps_ids is on {maps_ids.device}")
    if faces_verts_uvs.dim() != 3 or faces_verts_uvs.shape[1:] != (3, 2):
        raise ValueError("faces_verts_uvs must have shape (F, 3, 2)")
    N = fragments.pix_to_face.shape[0]
    if maps.dim() != 5 or maps.shape[0] != N:
        raise ValueError("with maps_ids, maps must have shape (N, M, H, W, C)")
    if maps.shape[1] < 2:
        raise ValueError("with maps_ids, maps needs at least two maps per mesh (the reference divides by M - 1)")
    if maps_ids.dim() != 2 or maps_ids.shape[0] != N:
        raise ValueError("Expected maps_ids to be of shape (N, F); got %r" % repr(tuple(maps_ids.shape)))  # textures.py:903-905
    if maps_ids.dtype != torch.int64:
        raise RuntimeError("sample_textures_uv: maps_ids must be int64")
    if faces_verts_uvs.dtype != torch.float32 or maps.dtype != torch.float32:
        raise RuntimeError("sample_textures_uv: faces_verts_uvs and maps must be float32")
    return _SampleUVMulti.apply(fragments.pix_to_face, fragments.bary_coords, faces_verts_uvs, maps, maps_ids.reshape(-1),
                                int(bool(align_corners)), _PAD[padding_mode], _MODE[sampling_mode])


class _SampleAtlas(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pix_to_face, bary, atlas):
        dev = atlas.device
        p2f, b, at = pix_to_face.contiguous(), bary.contiguous(), atlas.contiguous()
        F, R, _, C = at.shape
        lib = _lib.load()
        with torch.cuda.device(dev):
            out = torch.empty(tuple(p2f.shape) + (C,), dtype=torch.float32, device=dev)
            if out.numel():
                rc = lib.p3d_sample_atlas_forward(_C._ptr(p2f), _C._ptr(b), _C._ptr(at), p2f.numel(), F, R, C, _C._ptr(out),
                                                  _C._stream(dev))
                _lib.check(rc, "sample_textures_atlas")
        ctx.save_for_backward(p2f, b)
        ctx.atlas_shape = (F, R, C)
        return out

    @staticmethod
    def backward(ctx, grad_texels):
        p2f, b = ctx.saved_tensors
        F, R, C = ctx.atlas_shape
        dev = b.device
        g = grad_texels.contiguous()
        lib = _lib.load()
        with torch.cuda.device(dev):
            ga = torch.empty((F, R, R, C), dtype=torch.float32, device=dev)
            if ga.numel():
                rc = lib.p3d_sample_atlas_backward(_C._ptr(g), _C._ptr(p2f), _C._ptr(b), p2f.numel(), F, R, C, _C._ptr(ga),
                                                   _C._stream(dev))
                _lib.check(rc, "sample_textures_atlas_backward")
        return None, None, ga


def sample_textures_atlas(fragments, atlas_packed) -> torch.Tensor:
    """TexturesAtlas.sample_textures(fragments) for atlas_packed = atlas_packed() (F,R,R,C) -> texels (N,H,W,K,C): the
    cell of the face's R x R grid nearest to the barycentric sample (textures.py:565-612).  Differentiable with respect
    to the atlas, not to the barycentric coordinates."""
    _C._check_fragments("sample_textures_atlas", fragments.pix_to_face, bary_coords=fragments.bary_coords, atlas_packed=atlas_packed)
    if atlas_packed.dim() != 4 or atlas_packed.shape[1] != atlas_packed.shape[2] or atlas_packed.shape[1] < 1:
        raise ValueError("atlas_packed must have shape (F, R, R, C)")
    if atlas_packed.dtype != torch.float32 or fragments.bary_coords.dtype != torch.float32:
        raise RuntimeError("sample_textures_atlas: atlas_packed and bary_coords must be float32")
    if fragments.bary_coords.shape != tuple(fragments.pix_to_face.shape) + (3,):
        raise ValueError("bary_coords must have shape pix_to_face.shape + (3,)")
    if atlas_packed.shape[3] < 1:
        raise ValueError("atlas_packed needs at least one channel")
    return _SampleAtlas.apply(fragments.pix_to_face, fragments.bary_coords, atlas_packed)
