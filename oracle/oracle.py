"""ctypes front-end of the C oracle (TEST INFRASTRUCTURE, not product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
All functions take and return CPU torch tensors with the reference's shapes/dtypes.
"""
import ctypes
import importlib.util
import os
import sys

import torch

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.build_oracle()
        _lib = ctypes.CDLL(path)
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _f32(t):
    return t.detach().to(device="cpu", dtype=torch.float32).contiguous()


def _i64(t):
    return t.detach().to(device="cpu", dtype=torch.int64).contiguous()


def _i32(t):
    return t.detach().to(device="cpu", dtype=torch.int32).contiguous()


def rasterize_meshes_naive(face_verts, mesh_first, mesh_count, neighbor_idx, image_size, blur_radius, K,
                           perspective_correct, clip_barycentric_coords, cull_backfaces, cpu_order=False):
    fv, mf, mc = _f32(face_verts), _i64(mesh_first), _i64(mesh_count)
    nb = _i64(neighbor_idx) if neighbor_idx is not None else None
    H, W = image_size
    N = mf.shape[0]
    p2f = torch.empty((N, H, W, K), dtype=torch.int64)
    zbuf = torch.empty((N, H, W, K), dtype=torch.float32)
    bary = torch.empty((N, H, W, K, 3), dtype=torch.float32)
    dists = torch.empty((N, H, W, K), dtype=torch.float32)
    lib().orc_rasterize_meshes_naive(_p(fv), _p(mf), _p(mc), _p(nb) if nb is not None else None, N, H, W,
                                     ctypes.c_float(blur_radius), K, int(perspective_correct),
                                     int(clip_barycentric_coords), int(cull_backfaces), int(cpu_order), _p(p2f),
                                     _p(zbuf), _p(bary), _p(dists))
    return p2f, zbuf, bary, dists


def rasterize_meshes_backward(face_verts, pix_to_face, grad_zbuf, grad_bary, grad_dists, perspective_correct,
                              clip_barycentric_coords, cuda_semantics=True, acc64=True):
    fv, p2f = _f32(face_verts), _i64(pix_to_face)
    gz, gb, gd = _f32(grad_zbuf), _f32(grad_bary), _f32(grad_dists)
    N, H, W, K = p2f.shape
    F = fv.shape[0]
    out = torch.zeros((F, 3, 3), dtype=torch.float32)
    lib().orc_rasterize_meshes_backward(_p(fv), _p(p2f), _p(gz), _p(gb), _p(gd), ctypes.c_int64(F), N, H, W, K,
                                        int(perspective_correct), int(clip_barycentric_coords), int(cuda_semantics),
                                        int(acc64), _p(out))
    return out


def _coarse(kind, elems, aux, first, count, image_size, blur_radius, bin_size, M):
    H, W = image_size
    N = first.shape[0]
    BH, BW = 1 + (H - 1) // bin_size, 1 + (W - 1) // bin_size
    out = torch.empty((N, BH, BW, M), dtype=torch.int32)
    ovf = ctypes.c_int32(0)
    lib().orc_rasterize_coarse(kind, _p(elems), _p(aux) if aux is not None else None, _p(first), _p(count), N, H, W,
                               ctypes.c_float(blur_radius), bin_size, M, _p(out), ctypes.byref(ovf))
    return out, bool(ovf.value)


def rasterize_meshes_coarse(face_verts, mesh_first, mesh_count, image_size, blur_radius, bin_size, M):
    return _coarse(0, _f32(face_verts), None, _i64(mesh_first), _i64(mesh_count), image_size, blur_radius, bin_size, M)


def rasterize_points_coarse(points, first, count, image_size, radius, bin_size, M):
    return _coarse(1, _f32(points), _f32(radius), _i64(first), _i64(count), image_size, 0.0, bin_size, M)


def rasterize_points_naive(points, first, count, image_size, radius, K):
    pts, fi, ct, r = _f32(points), _i64(first), _i64(count), _f32(radius)
    H, W = image_size
    N = fi.shape[0]
    idx = torch.empty((N, H, W, K), dtype=torch.int32)
    zbuf = torch.empty((N, H, W, K), dtype=torch.float32)
    dists = torch.empty((N, H, W, K), dtype=torch.float32)
    lib().orc_rasterize_points_naive(_p(pts), _p(fi), _p(ct), _p(r), N, H, W, K, _p(idx), _p(zbuf), _p(dists))
    return idx, zbuf, dists


def rasterize_points_backward(points, idxs, grad_zbuf, grad_dists, acc64=True):
    pts, ix, gz, gd = _f32(points), _i32(idxs), _f32(grad_zbuf), _f32(grad_dists)
    N, H, W, K = ix.shape
    P = pts.shape[0]
    out = torch.zeros((P, 3), dtype=torch.float32)
    lib().orc_rasterize_points_backward(_p(pts), _p(ix), _p(gz), _p(gd), ctypes.c_int64(P), N, H, W, K, int(acc64),
                                        _p(out))
    return out


_MODES = {"alphacomposite": 0, "weightedsumnorm": 1, "weightedsum": 2}


def composite_forward(mode, features, alphas, points_idx):
    f, a, ix = _f32(features), _f32(alphas), _i64(points_idx)
    C, P = f.shape
    N, K, H, W = a.shape
    out = torch.zeros((N, C, H, W), dtype=torch.float32)
    lib().orc_composite_forward(_MODES[mode], _p(f), _p(a), _p(ix), N, C, ctypes.c_int64(P), K, H, W, _p(out))
    return out


def composite_backward(mode, grad_out, features, alphas, points_idx):
    g, f, a, ix = _f32(grad_out), _f32(features), _f32(alphas), _i64(points_idx)
    C, P = f.shape
    N, K, H, W = a.shape
    gf = torch.zeros((C, P), dtype=torch.float32)
    ga = torch.zeros((N, K, H, W), dtype=torch.float32)
    lib().orc_composite_backward(_MODES[mode], _p(g), _p(f), _p(a), _p(ix), N, C, ctypes.c_int64(P), K, H, W, _p(gf),
                                 _p(ga))
    return gf, ga


def interp_forward(pix_to_face, bary, face_attrs):
    p2f, b, fa = _i64(pix_to_face), _f32(bary), _f32(face_attrs)
    P = p2f.shape[0]
    F, _, D = fa.shape
    out = torch.zeros((P, D), dtype=torch.float32)
    lib().orc_interp_forward(_p(p2f), _p(b), _p(fa), ctypes.c_int64(P), ctypes.c_int64(F), D, _p(out))
    return out


def interp_backward(pix_to_face, bary, face_attrs, grad_pix_attrs):
    p2f, b, fa, g = _i64(pix_to_face), _f32(bary), _f32(face_attrs), _f32(grad_pix_attrs)
    P = p2f.shape[0]
    F, _, D = fa.shape
    gb = torch.zeros((P, 3), dtype=torch.float32)
    gf = torch.zeros((F, 3, D), dtype=torch.float32)
    lib().orc_interp_backward(_p(p2f), _p(b), _p(fa), _p(g), ctypes.c_int64(P), ctypes.c_int64(F), D, _p(gb), _p(gf))
    return gb, gf


def sigmoid_alpha_blend(dists, pix_to_face, sigma):
    d, p2f = _f32(dists), _i64(pix_to_face)
    N, H, W, K = p2f.shape
    out = torch.empty((N, H, W), dtype=torch.float32)
    lib().orc_sigmoid_alpha_blend(_p(d), _p(p2f), ctypes.c_float(sigma), ctypes.c_int64(N * H * W), K, _p(out))
    return out


def sigmoid_alpha_blend_backward(grad_alphas, alphas, dists, pix_to_face, sigma):
    ga, al, d, p2f = _f32(grad_alphas), _f32(alphas), _f32(dists), _i64(pix_to_face)
    N, H, W, K = p2f.shape
    out = torch.empty((N, H, W, K), dtype=torch.float32)
    lib().orc_sigmoid_alpha_blend_backward(_p(ga), _p(al), _p(d), _p(p2f), ctypes.c_float(sigma),
                                           ctypes.c_int64(N * H * W), K, _p(out))
    return out


def clip_faces(face_verts, mesh_first, planes, cull, z_clip_value, perspective_correct):
    """planes: [left, right, top, bottom, znear, zfar] with None for unused.  Returns a dict with the ClippedFaces
    fields (trimmed), plus counts (F_clipped, T3, T4)."""
    fv, mf = _f32(face_verts), _i64(mesh_first)
    F, N = fv.shape[0], mf.shape[0]
    mask = sum(1 << i for i, v in enumerate(planes) if v is not None)
    pl = torch.tensor([0.0 if v is None else float(v) for v in planes], dtype=torch.float32)
    out_fv = torch.zeros((2 * F + 1, 3, 3), dtype=torch.float32)
    first_c = torch.zeros((N,), dtype=torch.int64)
    count_c = torch.zeros((N,), dtype=torch.int64)
    c2u = torch.zeros((2 * F + 1,), dtype=torch.int64)
    conv = torch.zeros((2 * F + 1, 3, 3), dtype=torch.float32)
    conv_idx = torch.zeros((2 * F + 1,), dtype=torch.int64)
    nbr = torch.zeros((2 * F + 1,), dtype=torch.int64)
    counts = torch.zeros((4,), dtype=torch.int64)
    has_z = z_clip_value is not None
    lib().orc_clip_faces(_p(fv), ctypes.c_int64(F), _p(mf), N, _p(pl), mask, int(bool(cull)), int(has_z),
                         ctypes.c_float(z_clip_value if has_z else 0.0), int(bool(perspective_correct)), _p(out_fv),
                         _p(first_c), _p(count_c), _p(c2u), _p(conv), _p(conv_idx), _p(nbr), _p(counts))
    Fc, T3, T4 = (int(x) for x in counts[:3])
    T = T3 + 2 * T4
    return dict(face_verts=out_fv[:Fc], first=first_c, count=count_c, faces_clipped_to_unclipped_idx=c2u[:Fc],
                barycentric_conversion=conv[:T], faces_clipped_to_conversion_idx=conv_idx[:Fc],
                clipped_faces_neighbor_idx=nbr[:Fc], counts=(Fc, T3, T4))


def convert_clipped(pix_to_face_clipped, bary_clipped, c2u, conv, conv_idx):
    p2f, b = _i64(pix_to_face_clipped), _f32(bary_clipped)
    S = p2f.numel()
    p2f_u = torch.empty_like(p2f)
    b_u = torch.empty_like(b)
    has = conv is not None and conv.numel() > 0
    lib().orc_convert_clipped(_p(p2f), _p(b), _p(_i64(c2u)), _p(_f32(conv)) if has else None,
                              _p(_i64(conv_idx)) if has else None, ctypes.c_int64(S), _p(p2f_u), _p(b_u))
    return p2f_u, b_u


def _per_image(v, N):
    if isinstance(v, torch.Tensor):
        return _f32(v).reshape(-1).expand(N).contiguous() if v.numel() == 1 else _f32(v).reshape(N)
    return torch.full((N,), float(v), dtype=torch.float32)


def softmax_rgb_blend(colors, pix_to_face, dists, zbuf, sigma, gamma, background_color, znear=1.0, zfar=100.0):
    c, p2f, d, z = _f32(colors), _i64(pix_to_face), _f32(dists), _f32(zbuf)
    N, H, W, K = p2f.shape
    bg = _f32(torch.as_tensor(background_color, dtype=torch.float32))
    zn, zf = _per_image(znear, N), _per_image(zfar, N)
    out = torch.empty((N, H, W, 4), dtype=torch.float32)
    lib().orc_softmax_rgb_blend(_p(c), _p(p2f), _p(d), _p(z), ctypes.c_float(sigma), ctypes.c_float(gamma), _p(bg),
                                _p(zn), _p(zf), ctypes.c_int64(N * H * W), ctypes.c_int64(H * W), K, _p(out))
    return out


def softmax_rgb_blend_backward(grad_out, colors, pix_to_face, dists, zbuf, sigma, gamma, background_color, znear=1.0,
                               zfar=100.0):
    g, c, p2f, d, z = _f32(grad_out), _f32(colors), _i64(pix_to_face), _f32(dists), _f32(zbuf)
    N, H, W, K = p2f.shape
    bg = _f32(torch.as_tensor(background_color, dtype=torch.float32))
    zn, zf = _per_image(znear, N), _per_image(zfar, N)
    gc = torch.empty((N, H, W, K, 3), dtype=torch.float32)
    gd = torch.empty((N, H, W, K), dtype=torch.float32)
    gz = torch.empty((N, H, W, K), dtype=torch.float32)
    lib().orc_softmax_rgb_blend_backward(_p(g), _p(c), _p(p2f), _p(d), _p(z), ctypes.c_float(sigma),
                                         ctypes.c_float(gamma), _p(bg), _p(zn), _p(zf), ctypes.c_int64(N * H * W),
                                         ctypes.c_int64(H * W), K, _p(gc), _p(gd), _p(gz))
    return gc, gd, gz


# ---------------------------------------------------------------------------
# The real reference, when its CPU build is present (oracle/_ref/p3d_ref_cpu.so).
# ---------------------------------------------------------------------------
_ref_mod = None


def ref_module():
    """Return the reference's own CPU extension (or None when it was never built)."""
    global _ref_mod
    if _ref_mod is not None:
        return _ref_mod
    path = _build.build_ref() if _build.have_reference() else (_build.REF_SO if os.path.exists(_build.REF_SO) else None)
    if path is None or not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location("p3d_ref_cpu", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _ref_mod = mod
    return mod


def phong_shade(pix_to_face, bary, face_attrs, texels, params, point_light):
    """shading.py:59-112 restated (oracle/p3d_oracle.c: orc_phong_forward).  face_attrs (F,3,6|9), params (N,25)."""
    p2f, b, fa, pr = _i64(pix_to_face), _f32(bary), _f32(face_attrs), _f32(params)
    N, H, W, K = p2f.shape
    D = fa.shape[2]
    tx = _f32(texels) if texels is not None else None
    out = torch.zeros((N, H, W, K, 3), dtype=torch.float32)
    lib().orc_phong_forward(_p(p2f), _p(b), _p(fa), D, _p(tx) if tx is not None else None, _p(pr), int(point_light), N,
                            ctypes.c_int64(H * W * K), _p(out))
    return out


def phong_shade_backward(grad_colors, pix_to_face, bary, face_attrs, texels, params, point_light, with_params=False):
    g, p2f, b, fa, pr = _f32(grad_colors), _i64(pix_to_face), _f32(bary), _f32(face_attrs), _f32(params)
    N, H, W, K = p2f.shape
    F, _, D = fa.shape
    tx = _f32(texels) if texels is not None else None
    gb = torch.zeros((N, H, W, K, 3), dtype=torch.float32)
    gf = torch.zeros((F, 3, D), dtype=torch.float32)
    gt = torch.zeros((N, H, W, K, 3), dtype=torch.float32) if D == 6 else None
    gp = torch.zeros((N, 25), dtype=torch.float32)
    lib().orc_phong_backward(_p(g), _p(p2f), _p(b), _p(fa), D, _p(tx) if tx is not None else None, _p(pr),
                             int(point_light), N, ctypes.c_int64(H * W * K), ctypes.c_int64(F), _p(gb), _p(gf),
                             _p(gt) if gt is not None else None, _p(gp))
    return (gb, gf, gt, gp) if with_params else (gb, gf, gt)


_PAD = {"zeros": 0, "border": 1}
_SMODE = {"bilinear": 0, "nearest": 1}


def sample_uv(pix_to_face, bary, face_uvs, maps, align_corners=True, padding_mode="border", sampling_mode="bilinear"):
    """TexturesUV.sample_textures restated (oracle/p3d_oracle.c: orc_sample_uv_forward)."""
    p2f, b, fu, mp = _i64(pix_to_face), _f32(bary), _f32(face_uvs), _f32(maps)
    N, H, W, K = p2f.shape
    _, Hm, Wm, C = mp.shape
    out = torch.zeros((N, H, W, K, C), dtype=torch.float32)
    lib().orc_sample_uv_forward(_p(p2f), _p(b), _p(fu), _p(mp), N, ctypes.c_int64(H * W * K), Hm, Wm, C, int(align_corners),
                                _PAD[padding_mode], _SMODE[sampling_mode], _p(out))
    return out


def sample_uv_backward(grad_texels, pix_to_face, bary, face_uvs, maps, align_corners=True, padding_mode="border",
                       sampling_mode="bilinear"):
    g, p2f, b, fu, mp = _f32(grad_texels), _i64(pix_to_face), _f32(bary), _f32(face_uvs), _f32(maps)
    N, H, W, K = p2f.shape
    _, Hm, Wm, C = mp.shape
    F = fu.shape[0]
    gb = torch.zeros((N, H, W, K, 3), dtype=torch.float32)
    gfu = torch.zeros((F, 3, 2), dtype=torch.float32)
    gm = torch.zeros_like(mp)
    lib().orc_sample_uv_backward(_p(g), _p(p2f), _p(b), _p(fu), _p(mp), N, ctypes.c_int64(H * W * K), ctypes.c_int64(F), Hm,
                                 Wm, C, int(align_corners), _PAD[padding_mode], _SMODE[sampling_mode], _p(gb), _p(gfu), _p(gm))
    return gb, gfu, gm


def sample_uv_multi(pix_to_face, bary, face_uvs, maps, maps_ids, align_corners=True, padding_mode="border",
                    sampling_mode="bilinear"):
    """TexturesUV.sample_textures with maps_ids restated (oracle/p3d_oracle.c: orc_sample_uv_multi_forward).  maps
    (N,M,Hm,Wm,C); maps_ids: maps_ids_padded (N,Fmax), flattened as the reference does."""
    p2f, b, fu, mp, ids = _i64(pix_to_face), _f32(bary), _f32(face_uvs), _f32(maps), _i64(maps_ids).reshape(-1)
    N, H, W, K = p2f.shape
    _, M, Hm, Wm, C = mp.shape
    out = torch.zeros((N, H, W, K, C), dtype=torch.float32)
    lib().orc_sample_uv_multi_forward(_p(p2f), _p(b), _p(fu), _p(mp), _p(ids), ctypes.c_int64(ids.numel()), N,
                                      ctypes.c_int64(H * W * K), M, Hm, Wm, C, int(align_corners), _PAD[padding_mode],
                                      _SMODE[sampling_mode], _p(out))
    return out


def sample_uv_multi_backward(grad_texels, pix_to_face, bary, face_uvs, maps, maps_ids, align_corners=True,
                             padding_mode="border", sampling_mode="bilinear"):
    g, p2f, b, fu, mp = _f32(grad_texels), _i64(pix_to_face), _f32(bary), _f32(face_uvs), _f32(maps)
    ids = _i64(maps_ids).reshape(-1)
    N, H, W, K = p2f.shape
    _, M, Hm, Wm, C = mp.shape
    F = fu.shape[0]
    gb = torch.zeros((N, H, W, K, 3), dtype=torch.float32)
    gfu = torch.zeros((F, 3, 2), dtype=torch.float32)
    gm = torch.zeros_like(mp)
    lib().orc_sample_uv_multi_backward(_p(g), _p(p2f), _p(b), _p(fu), _p(mp), _p(ids), ctypes.c_int64(ids.numel()), N,
                                       ctypes.c_int64(H * W * K), ctypes.c_int64(F), M, Hm, Wm, C, int(align_corners),
                                       _PAD[padding_mode], _SMODE[sampling_mode], _p(gb), _p(gfu), _p(gm))
    return gb, gfu, gm


def sample_atlas(pix_to_face, bary, atlas):
    """TexturesAtlas.sample_textures restated (oracle/p3d_oracle.c: orc_sample_atlas_forward)."""
    p2f, b, at = _i64(pix_to_face), _f32(bary), _f32(atlas)
    F, R, _, C = at.shape
    out = torch.zeros(tuple(p2f.shape) + (C,), dtype=torch.float32)
    lib().orc_sample_atlas_forward(_p(p2f), _p(b), _p(at), ctypes.c_int64(p2f.numel()), ctypes.c_int64(F), R, C, _p(out))
    return out


def sample_atlas_backward(grad_texels, pix_to_face, bary, atlas_shape):
    g, p2f, b = _f32(grad_texels), _i64(pix_to_face), _f32(bary)
    F, R, _, C = atlas_shape
    ga = torch.zeros((F, R, R, C), dtype=torch.float32)
    lib().orc_sample_atlas_backward(_p(g), _p(p2f), _p(b), ctypes.c_int64(p2f.numel()), ctypes.c_int64(F), R, C, _p(ga))
    return ga


def hard_rgb_blend(colors, pix_to_face, background):
    """hard_rgb_blend restated (oracle/p3d_oracle.c: orc_hard_rgb_blend_forward)."""
    c, p2f = _f32(colors), _i64(pix_to_face)
    N, H, W, K = p2f.shape
    bg = _f32(torch.as_tensor(background, dtype=torch.float32))
    out = torch.zeros((N, H, W, 4), dtype=torch.float32)
    lib().orc_hard_rgb_blend_forward(_p(c), _p(p2f), _p(bg), ctypes.c_int64(N * H * W), K, _p(out))
    return out


def hard_rgb_blend_backward(grad_out, pix_to_face):
    g, p2f = _f32(grad_out), _i64(pix_to_face)
    N, H, W, K = p2f.shape
    gc = torch.zeros((N, H, W, K, 3), dtype=torch.float32)
    lib().orc_hard_rgb_blend_backward(_p(g), _p(p2f), ctypes.c_int64(N * H * W), K, _p(gc))
    return gc


_ref_hip = {}


def ref_hip_module(nofma=False):
    """The reference's own CUDA kernels built for gfx950 (oracle/build_ref_hip.py), or None.  nofma: the variant compiled
    with -ffp-contract=off (the reference's expression order itself); default: hipcc's defaults, like a user's build."""
    key = bool(nofma)
    if key in _ref_hip:
        return _ref_hip[key]
    name = "p3d_ref_hip_nofma" if nofma else "p3d_ref_hip"
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", name + ".so")
    mod = None
    if os.path.exists(path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    _ref_hip[key] = mod
    return mod
