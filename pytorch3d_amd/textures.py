"""Host-side mirror of the texture classes' sample_textures over the C ABI.

TexturesAtlas.sample_textures (pytorch3d/renderer/mesh/textures.py:565-612): `sample_textures_atlas`, one kernel each way
(include/p3d_amd.h: p3d_sample_atlas_forward / _backward), gradient to the atlas only -- as in the reference.

TexturesUV.sample_textures (textures.py:1190-1268): uv interpolation + grid_sample fused into one kernel each way (include/p3d_amd.h:
p3d_sample_uv_forward / _backward).  One texture map per mesh (`maps_ids` is not provided); padding "zeros" or
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


def sample_textures_uv(fragments, faces_verts_uvs, maps, align_corners: bool = True, padding_mode: str = "border",
                       sampling_mode: str = "bilinear") -> torch.Tensor:
    """TexturesUV.sample_textures(fragments) for faces_verts_uvs = cat(verts_uvs_list[i][faces_uvs_list[i]]) (F,3,2) and
    maps = maps_padded() (N,Hm,Wm,C) -> texels (N,H,W,K,C).  Defaults as TexturesUV.__init__ (textures.py:716-718)."""
    if padding_mode not in _PAD:
        raise NotImplementedError(f"sample_textures_uv: padding_mode {padding_mode!r} (only 'zeros' and 'border')")
    if sampling_mode not in _MODE:
        raise ValueError(f"sample_textures_uv: sampling_mode {sampling_mode!r}")
    for name, t in (("pix_to_face", fragments.pix_to_face), ("bary_coords", fragments.bary_coords),
                    ("faces_verts_uvs", faces_verts_uvs), ("maps", maps)):
        _C._need_gpu(t, name)
    if faces_verts_uvs.dim() != 3 or faces_verts_uvs.shape[1:] != (3, 2):
        raise ValueError("faces_verts_uvs must have shape (F, 3, 2)")
    if maps.dim() != 4 or maps.shape[0] != fragments.pix_to_face.shape[0]:
        raise ValueError("maps must have shape (N, H, W, C) with one map per batch element")
    if faces_verts_uvs.dtype != torch.float32 or maps.dtype != torch.float32:
        raise RuntimeError("sample_textures_uv: faces_verts_uvs and maps must be float32")
    return _SampleUV.apply(fragments.pix_to_face, fragments.bary_coords, faces_verts_uvs, maps, int(bool(align_corners)),
                           _PAD[padding_mode], _MODE[sampling_mode])


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
    for name, t in (("pix_to_face", fragments.pix_to_face), ("bary_coords", fragments.bary_coords),
                    ("atlas_packed", atlas_packed)):
        _C._need_gpu(t, name)
    if atlas_packed.dim() != 4 or atlas_packed.shape[1] != atlas_packed.shape[2] or atlas_packed.shape[1] < 1:
        raise ValueError("atlas_packed must have shape (F, R, R, C)")
    if atlas_packed.dtype != torch.float32 or fragments.bary_coords.dtype != torch.float32:
        raise RuntimeError("sample_textures_atlas: atlas_packed and bary_coords must be float32")
    if fragments.bary_coords.shape != tuple(fragments.pix_to_face.shape) + (3,):
        raise ValueError("bary_coords must have shape pix_to_face.shape + (3,)")
    if atlas_packed.shape[3] < 1:
        raise ValueError("atlas_packed needs at least one channel")
    return _SampleAtlas.apply(fragments.pix_to_face, fragments.bary_coords, atlas_packed)
