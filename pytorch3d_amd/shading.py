"""Host-side mirror of pytorch3d/renderer/mesh/shading.py:17-112 over the C ABI.

`phong_shading(meshes, fragments, lights, cameras, materials, texels)` has the reference's signature and return
value.  The reference runs two `interpolate_face_attributes` calls, `lights.diffuse`, `lights.specular`
(renderer/lighting.py:17-159) and the colour mix as ~45 torch kernels over (N,H,W,K,3) tensors plus their autograd
twins; here it is ONE kernel forward and ONE backward (include/p3d_amd.h: p3d_phong_shade_forward / _backward).
`phong_shading_vertex_colors` additionally fuses the `TexturesVertex` texel interpolation
(renderer/mesh/textures.py: sample_textures = interpolate_face_attributes of the per-face vertex colours).
`flat_shading` (shading.py:178-225) runs on the same kernels with per-face records (face centre, face normal);
`gouraud_shading` (shading.py:125-175) lights the V vertices with plain torch ops (V-sized, not the hot part) and
interpolates the shaded colours with the fused interpolate_face_attributes kernels.

`lights`, `cameras`, `materials` are duck-typed: any object with the reference's attribute names works
(`PointLights.location`, `DirectionalLights.direction`, `*.ambient_color / diffuse_color / specular_color`,
`Materials.shininess`, `cameras.get_camera_center()`), each (1, 3) or (N, 3).  Gradients flow to the mesh
vertices, vertex normals, texels / vertex colours and barycentric coordinates, and -- when one of them requires grad --
to the lights, materials and camera centre (the backward kernel then also reduces the gradient of the (N, 25)
parameter block, which torch's autograd distributes over the tensors it was packed from).
"""
from typing import NamedTuple, Optional

import torch
import torch.nn.functional as Fn

from . import _C, _lib
from .interp_face_attrs import interpolate_face_attributes
from .rasterize_meshes import gather_face_verts

PARAM_FLOATS = 25  # include/p3d_amd.h: P3D_SHADE_PARAM_FLOATS
LIGHT_DIRECTIONAL, LIGHT_POINT = 0, 1


class Lights(NamedTuple):
    """Minimal stand-in for renderer/lighting.py's PointLights / DirectionalLights / AmbientLights (out of scope,
    used unmodified when the reference is installed): set `location` for a point light, `direction` for a
    directional one, neither for ambient-only."""

    ambient_color: torch.Tensor
    diffuse_color: Optional[torch.Tensor] = None
    specular_color: Optional[torch.Tensor] = None
    location: Optional[torch.Tensor] = None
    direction: Optional[torch.Tensor] = None


class Materials(NamedTuple):
    """renderer/materials.py:15-60."""

    ambient_color: torch.Tensor
    diffuse_color: torch.Tensor
    specular_color: torch.Tensor
    shininess: torch.Tensor


def _rows(v, N, C, device, name):
    t = v if torch.is_tensor(v) else torch.tensor(v, dtype=torch.float32)
    t = t.to(device=device, dtype=torch.float32)  # differentiable: the kernels return the gradient of the packed block
    if C == 1:
        t = t.reshape(-1, 1)
    if t.ndim == 1:
        t = t[None]
    if t.ndim != 2 or t.shape[1] != C or t.shape[0] not in (1, N):
        raise ValueError(f"phong_shading: {name} must have shape (1, {C}) or ({N}, {C}); got {tuple(t.shape)}")
    return t.expand(N, C)


def pack_shade_params(lights, cameras, materials, N, device):
    """-> ((N, 25) float32 parameter block, light kind) as p3d_phong_shade_* take them."""
    if getattr(lights, "location", None) is not None:
        kind, vec = LIGHT_POINT, lights.location
    elif getattr(lights, "direction", None) is not None:
        kind, vec = LIGHT_DIRECTIONAL, lights.direction
    else:  # AmbientLights: diffuse() and specular() return zeros (lighting.py:339-352)
        kind, vec = LIGHT_DIRECTIONAL, ((0.0, 0.0, 0.0),)
    zero = ((0.0, 0.0, 0.0),)
    ldiff = getattr(lights, "diffuse_color", None)
    lspec = getattr(lights, "specular_color", None)
    if getattr(lights, "location", None) is None and getattr(lights, "direction", None) is None:
        ldiff = lspec = None
    cols = [
        _rows(lights.ambient_color, N, 3, device, "lights.ambient_color"),
        _rows(ldiff if ldiff is not None else zero, N, 3, device, "lights.diffuse_color"),
        _rows(lspec if lspec is not None else zero, N, 3, device, "lights.specular_color"),
        _rows(vec, N, 3, device, "lights.location / direction"),
        _rows(materials.ambient_color, N, 3, device, "materials.ambient_color"),
        _rows(materials.diffuse_color, N, 3, device, "materials.diffuse_color"),
        _rows(materials.specular_color, N, 3, device, "materials.specular_color"),
        _rows(materials.shininess, N, 1, device, "materials.shininess"),
        _rows(cameras.get_camera_center(), N, 3, device, "cameras.get_camera_center()"),
    ]
    return torch.cat(cols, 1), kind


class _PhongShade(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pix_to_face, bary, face_attrs, texels, params, kind):
        N, H, W, K = pix_to_face.shape
        dev = bary.device
        p2f, b, fa = pix_to_face.contiguous(), bary.contiguous(), face_attrs.contiguous()
        tx = texels.contiguous() if texels is not None else None
        params = params.contiguous()
        F, _, D = fa.shape
        lib = _lib.load()
        with torch.cuda.device(dev):
            colors = torch.empty((N, H, W, K, 3), dtype=torch.float32, device=dev)
            if colors.numel():
                rc = lib.p3d_phong_shade_forward(_C._ptr(p2f), _C._ptr(b), _C._ptr(fa), D, _C._ptr(tx), _C._ptr(params),
                                                 kind, N, H, W, K, F, _C._ptr(colors), _C._stream(dev))
                _lib.check(rc, "phong_shading")
        ctx.save_for_backward(p2f, b, fa, tx if tx is not None else torch.empty(0, device=dev), params)
        ctx.kind = kind
        ctx.has_texels = tx is not None
        return colors

    @staticmethod
    def backward(ctx, grad_colors):
        _refuse_when_deterministic("phong_shading backward")
        p2f, b, fa, tx, params = ctx.saved_tensors
        N, H, W, K = p2f.shape
        F, _, D = fa.shape
        dev = b.device
        g = grad_colors.contiguous()
        lib = _lib.load()
        with torch.cuda.device(dev):
            gb = torch.empty((N, H, W, K, 3), dtype=torch.float32, device=dev)
            gfa = torch.empty((F, 3, D), dtype=torch.float32, device=dev)
            gt = torch.empty((N, H, W, K, 3), dtype=torch.float32, device=dev) if ctx.has_texels else None
            gp = torch.empty((N, PARAM_FLOATS), dtype=torch.float32, device=dev) if ctx.needs_input_grad[4] else None
            rc = lib.p3d_phong_shade_backward(_C._ptr(g), _C._ptr(p2f), _C._ptr(b), _C._ptr(fa), D,
                                              _C._ptr(tx if ctx.has_texels else None), _C._ptr(params), ctx.kind, N, H, W,
                                              K, F, _C._ptr(gb), _C._ptr(gfa), _C._ptr(gt), _C._ptr(gp), _C._stream(dev))
            _lib.check(rc, "phong_shading_backward")
        return None, gb, gfa, gt, gp, None


def _check(fragments, *named):
    _C._need_gpu(fragments.pix_to_face, "pix_to_face")
    _C._need_gpu(fragments.bary_coords, "bary_coords")
    for name, t in named:
        _C._need_gpu(t, name)
        if t.dtype != torch.float32:
            raise RuntimeError(f"phong_shading: {name} must be float32")


def _face_records(meshes, extra=None):
    """(F, 3, 6|9): per face corner [vertex xyz | vertex normal | extra], differentiable (shading.py:84-88)."""
    faces = meshes.faces_packed()
    fv = gather_face_verts(meshes.verts_packed(), faces)
    fn = gather_face_verts(meshes.verts_normals_packed(), faces)
    parts = [fv, fn] + ([extra] if extra is not None else [])
    return torch.cat(parts, 2)


def phong_shading(meshes, fragments, lights, cameras, materials, texels) -> torch.Tensor:
    """shading.py:99-112.  texels (N,H,W,K,3) -> colors (N,H,W,K,3)."""
    _check(fragments, ("texels", texels))
    N = fragments.pix_to_face.shape[0]
    params, kind = pack_shade_params(lights, cameras, materials, N, texels.device)
    return _PhongShade.apply(fragments.pix_to_face, fragments.bary_coords, _face_records(meshes), texels, params, kind)


def phong_shading_vertex_colors(meshes, fragments, lights, cameras, materials, verts_colors_packed) -> torch.Tensor:
    """phong_shading(..., texels=TexturesVertex(verts_colors).sample_textures(fragments)) with the texel interpolation
    (textures.py sample_textures -> interpolate_face_attributes) fused into the same kernel.  verts_colors_packed (V,3)."""
    _check(fragments, ("verts_colors_packed", verts_colors_packed))
    N = fragments.pix_to_face.shape[0]
    params, kind = pack_shade_params(lights, cameras, materials, N, verts_colors_packed.device)
    fc = gather_face_verts(verts_colors_packed, meshes.faces_packed())
    return _PhongShade.apply(fragments.pix_to_face, fragments.bary_coords, _face_records(meshes, fc), None, params, kind)


def flat_shading(meshes, fragments, lights, cameras, materials, texels) -> torch.Tensor:
    """shading.py:178-225: lighting at the face centre with the face normal, same for every pixel of the face.  Runs the
    Phong kernels on records whose three corners are identical; the barycentric coordinates (which sum to 1) then only
    select the face, and get no gradient -- as in the reference, where pixels gather per-face values by index."""
    _check(fragments, ("texels", texels))
    N = fragments.pix_to_face.shape[0]
    params, kind = pack_shade_params(lights, cameras, materials, N, texels.device)
    fv = gather_face_verts(meshes.verts_packed(), meshes.faces_packed())
    rec = torch.cat([fv.mean(dim=-2), meshes.faces_normals_packed()], 1)  # (F, 6): face centre | face normal
    rec = rec[:, None, :].expand(rec.shape[0], 3, 6)
    return _PhongShade.apply(fragments.pix_to_face, fragments.bary_coords.detach(), rec, texels, params, kind)


def _light_points(points, normals, rows, kind):
    """_apply_lighting (shading.py:17-56, lighting.py:17-159) for packed (P,3) points with one parameter row (P,25) each,
    as torch ops: the V-sized vertex stage of Gouraud shading."""
    la, ld, ls, vec = rows[:, 0:3], rows[:, 3:6], rows[:, 6:9], rows[:, 9:12]
    ma, md, ms, shin, cam = rows[:, 12:15], rows[:, 15:18], rows[:, 18:21], rows[:, 21], rows[:, 22:25]
    direction = vec - points if kind == LIGHT_POINT else vec
    n = Fn.normalize(normals, p=2, dim=-1, eps=1e-6)
    d = Fn.normalize(direction, p=2, dim=-1, eps=1e-6)
    cos = (n * d).sum(-1)
    light_diffuse = ld * torch.relu(cos)[:, None]
    view = Fn.normalize(cam - points, p=2, dim=-1, eps=1e-6)
    reflect = -d + 2 * (cos[:, None] * n)
    alpha = torch.relu((view * reflect).sum(-1)) * (cos > 0).to(torch.float32)
    light_specular = ls * torch.pow(alpha, shin)[:, None]
    return ma * la, md * light_diffuse, ms * light_specular


def gouraud_shading(meshes, fragments, lights, cameras, materials, verts_colors_packed=None) -> torch.Tensor:
    """shading.py:125-175: light the vertices, interpolate the shaded vertex colours.  verts_colors_packed defaults to
    meshes.textures.verts_features_packed() (the reference accepts TexturesVertex only)."""
    if verts_colors_packed is None:
        tex = getattr(meshes, "textures", None)
        if tex is None or not hasattr(tex, "verts_features_packed"):
            raise ValueError("Mesh textures must be an instance of TexturesVertex")
        verts_colors_packed = tex.verts_features_packed()
    _check(fragments, ("verts_colors_packed", verts_colors_packed))
    verts, faces = meshes.verts_packed(), meshes.faces_packed()
    N = fragments.pix_to_face.shape[0]
    params, kind = pack_shade_params(lights, cameras, materials, N, verts.device)
    rows = params[meshes.verts_packed_to_mesh_idx()]  # gather_props(vert_to_mesh_idx), shading.py:158-161
    ambient, diffuse, specular = _light_points(verts, meshes.verts_normals_packed(), rows, kind)
    shaded = verts_colors_packed * (ambient + diffuse) + specular
    return interpolate_face_attributes(fragments.pix_to_face, fragments.bary_coords, gather_face_verts(shaded, faces))


# ---------------------------------------------------------------------------------------------------------------------
# SoftPhongShader in one kernel each way (csrc/soft_phong.hip): phong_shading + softmax_rgb_blend, the colours never in HBM
# ---------------------------------------------------------------------------------------------------------------------
def _refuse_when_deterministic(what):
    """The shading backward kernels scatter to the face records (and light / material parameters) with float atomics: run to
    run the sums differ in the last bits.  Under torch.use_deterministic_algorithms(True) that has to be said, as the
    rasterizer's backward says it (rasterize_meshes.cu:587 alertNotDeterministic)."""
    if torch.are_deterministic_algorithms_enabled() and not torch.is_deterministic_algorithms_warn_only_enabled():
        raise RuntimeError(f"{what} does not have a deterministic implementation (float atomics); use the unfused torch "
                           "formulation of the reference or torch.use_deterministic_algorithms(True, warn_only=True)")


class _SoftPhong(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pix_to_face, bary, dists, zbuf, face_attrs, texels, params, kind, sigma, gamma, bg, znear, zfar):
        from .blending import _plane

        N, H, W, K = pix_to_face.shape
        dev = bary.device
        p2f, b, d, z = pix_to_face.contiguous(), bary.contiguous(), dists.contiguous(), zbuf.contiguous()
        fa = face_attrs.contiguous()
        tx = texels.contiguous() if texels is not None else None
        params = params.contiguous()
        F, _, D = fa.shape
        zn, zn_t = _plane(znear, N, dev)
        zf, zf_t = _plane(zfar, N, dev)
        lib = _lib.load()
        with torch.cuda.device(dev):
            out = torch.empty((N, H, W, 4), dtype=torch.float32, device=dev)
            if out.numel():
                rc = lib.p3d_soft_phong_forward(_C._ptr(p2f), _C._ptr(b), _C._ptr(d), _C._ptr(z), _C._ptr(fa), D, _C._ptr(tx),
                                                _C._ptr(params), kind, float(sigma), float(gamma), bg, zn, zf, _C._ptr(zn_t),
                                                _C._ptr(zf_t), N, H, W, K, F, _C._ptr(out), _C._stream(dev))
                _lib.check(rc, "soft_phong_shading")
        empty = torch.empty(0, device=dev)
        ctx.save_for_backward(p2f, b, d, z, fa, tx if tx is not None else empty, params, zn_t if zn_t is not None else empty,
                              zf_t if zf_t is not None else empty)
        ctx.meta = (kind, float(sigma), float(gamma), bg, zn, zf, tx is not None, zn_t is not None, zf_t is not None)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        _refuse_when_deterministic("soft_phong_shading backward")
        p2f, b, d, z, fa, tx, params, zn_t, zf_t = ctx.saved_tensors
        kind, sigma, gamma, bg, zn, zf, has_tx, has_zn, has_zf = ctx.meta
        N, H, W, K = p2f.shape
        F, _, D = fa.shape
        dev = b.device
        g = grad_out.contiguous()
        lib = _lib.load()
        with torch.cuda.device(dev):
            gb = torch.empty((N, H, W, K, 3), dtype=torch.float32, device=dev)
            gd = torch.empty((N, H, W, K), dtype=torch.float32, device=dev)
            gz = torch.empty((N, H, W, K), dtype=torch.float32, device=dev)
            gfa = torch.empty((F, 3, D), dtype=torch.float32, device=dev)
            gt = torch.empty((N, H, W, K, 3), dtype=torch.float32, device=dev) if has_tx else None
            gp = torch.empty((N, PARAM_FLOATS), dtype=torch.float32, device=dev) if ctx.needs_input_grad[6] else None
            rc = lib.p3d_soft_phong_backward(_C._ptr(g), _C._ptr(p2f), _C._ptr(b), _C._ptr(d), _C._ptr(z), _C._ptr(fa), D,
                                             _C._ptr(tx if has_tx else None), _C._ptr(params), kind, sigma, gamma, bg, zn, zf,
                                             _C._ptr(zn_t if has_zn else None), _C._ptr(zf_t if has_zf else None), N, H, W, K, F,
                                             _C._ptr(gb), _C._ptr(gd), _C._ptr(gz), _C._ptr(gfa), _C._ptr(gt), _C._ptr(gp),
                                             _C._stream(dev))
            _lib.check(rc, "soft_phong_shading_backward")
        return None, gb, gd, gz, gfa, gt, gp, None, None, None, None, None, None


def soft_phong_supported(fragments) -> bool:
    """The fused kernels take K in {1, 2, 4, 8, 16} (a pixel's slots are adjacent lanes of one DPP row)."""
    return int(fragments.pix_to_face.shape[-1]) in (1, 2, 4, 8, 16)


def soft_phong_shading(meshes, fragments, lights, cameras, materials, texels, blend_params, znear=1.0, zfar=100.0,
                       verts_colors_packed=None) -> torch.Tensor:
    """SoftPhongShader.forward (renderer/mesh/shader.py:113-147) without its intermediate:
    `softmax_rgb_blend(phong_shading(meshes, fragments, lights, cameras, materials, texels), fragments, blend_params,
    znear, zfar)` -> RGBA (N,H,W,4), one kernel forward and one backward.  texels (N,H,W,K,3), or None with
    verts_colors_packed (V,3): the TexturesVertex interpolation is fused in as well.  K outside {1, 2, 4, 8, 16}: the two
    operators run one after the other."""
    from .blending import _background, softmax_rgb_blend

    if not soft_phong_supported(fragments):
        colors = (phong_shading(meshes, fragments, lights, cameras, materials, texels) if texels is not None else
                  phong_shading_vertex_colors(meshes, fragments, lights, cameras, materials, verts_colors_packed))
        return softmax_rgb_blend(colors, fragments, blend_params, znear=znear, zfar=zfar)
    if (texels is None) == (verts_colors_packed is None):
        raise ValueError("soft_phong_shading: give either texels or verts_colors_packed")
    named = [("dists", fragments.dists), ("zbuf", fragments.zbuf)]
    named.append(("texels", texels) if texels is not None else ("verts_colors_packed", verts_colors_packed))
    _check(fragments, *named)
    if fragments.pix_to_face.dtype != torch.int64:
        raise RuntimeError("soft_phong_shading: pix_to_face must be int64")
    N = fragments.pix_to_face.shape[0]
    dev = fragments.bary_coords.device
    params, kind = pack_shade_params(lights, cameras, materials, N, dev)
    if texels is not None:
        rec = _face_records(meshes)
    else:
        rec = _face_records(meshes, gather_face_verts(verts_colors_packed, meshes.faces_packed()))
    bg = _background(blend_params, dev, "soft_phong_shading")
    return _SoftPhong.apply(fragments.pix_to_face, fragments.bary_coords, fragments.dists, fragments.zbuf, rec, texels, params, kind,
                            blend_params.sigma, blend_params.gamma, bg, znear, zfar)
