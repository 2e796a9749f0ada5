"""GPU parity of the shading row: the fused Phong kernels (p3d_phong_shade_forward / _backward)
against the reference-generated fixture (tests/golden/shading_ref.npz: the reference's phong_shading + lighting
classes + torch autograd), the oracle on random inputs (every K path, ragged images, shininess 0 / 1, per-image
materials), and -- on bench-generator fragments at 512x512 -- shading.py:59-112 / lighting.py:17-159 restated with
torch ops on the GPU (+ torch autograd).  Tolerances: colours 1e-5 (north_star), gradients rtol 1e-3.
"""
from collections import namedtuple

import pytest
import torch

import _util as U
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
Frag = namedtuple("Frag", "pix_to_face bary_coords")


class MeshView:
    """Exactly what phong_shading reads from a Meshes object."""

    def __init__(self, verts, faces, normals):
        self.v, self.f, self.n = verts, faces, normals

    def verts_packed(self):
        return self.v

    def faces_packed(self):
        return self.f

    def verts_normals_packed(self):
        return self.n


class Cam:
    def __init__(self, c):
        self.c = c

    def get_camera_center(self):
        return self.c


def _close(a, b, rtol=1e-3, atol=2e-5):
    return torch.allclose(a, b, rtol=rtol, atol=atol * max(1.0, b.abs().max().item()))


def _lights_of(g, tag, d):
    import pytorch3d_amd.shading as sh

    get = lambda n: g[f"{tag}_light_{n}"].to(d) if f"{tag}_light_{n}" in g else None
    L = sh.Lights(ambient_color=get("ambient_color"), diffuse_color=get("diffuse_color"),
                  specular_color=get("specular_color"), location=get("location"), direction=get("direction"))
    M = sh.Materials(*(g[f"{tag}_mat_{n}"].to(d) for n in ("ambient_color", "diffuse_color", "specular_color", "shininess")))
    return L, M


@pytest.mark.parametrize("tag", ["point", "dir", "amb"])
@pytest.mark.parametrize("kind", ["texels", "vcol"])
def test_phong_shading_mirror_vs_reference_fixture(tag, kind):
    import pytorch3d_amd as p3d

    g = U.shading_golden()
    d = torch.device("cuda:0")
    pre = f"{tag}_{kind}_"
    v = g["verts"].to(d).requires_grad_(True)
    nrm = g["normals"].to(d).requires_grad_(True)
    b = g["bary"].to(d).requires_grad_(True)
    L, M = _lights_of(g, tag, d)
    mesh, frag, cam = MeshView(v, g["faces"].to(d), nrm), Frag(g["pix_to_face"].to(d), b), Cam(g["camera_center"].to(d))
    if kind == "texels":
        t_in = g["texels"].to(d).requires_grad_(True)
        col = p3d.phong_shading(mesh, frag, L, cam, M, t_in)
    else:
        t_in = g["verts_colors"].to(d).requires_grad_(True)
        col = p3d.phong_shading_vertex_colors(mesh, frag, L, cam, M, t_in)
    ref = g[pre + "colors"]
    assert torch.allclose(col.cpu(), ref, atol=1e-5, rtol=1e-5), (col.cpu() - ref).abs().max()
    col.backward(g[pre + "grad_colors"].to(d))
    for got, name in ((v.grad, "grad_verts"), (nrm.grad, "grad_normals"), (b.grad, "grad_bary"), (t_in.grad, "grad_tex")):
        if pre + name in g:
            assert _close(got.cpu(), g[pre + name]), name
        elif name != "grad_bary":  # ambient-only lights: the reference's graph does not reach the geometry
            assert got is None or got.abs().max() == 0, name


@pytest.mark.parametrize("tag", ["point", "dir"])
def test_light_material_camera_gradients_vs_reference_autograd(tag):
    """Lights, materials and the camera centre that require grad: the backward kernel reduces the gradient of the packed
    (N,25) block, torch's autograd distributes it over the tensors it was packed from."""
    import pytorch3d_amd as p3d
    import pytorch3d_amd.shading as sh

    g = U.shading_golden()
    d = torch.device("cuda:0")
    vec = "location" if tag == "point" else "direction"
    leaf = lambda k: g[k].to(d).requires_grad_(True)
    lt = {n: leaf(f"{tag}_light_{n}") for n in ("ambient_color", "diffuse_color", "specular_color", vec)}
    mt = {n: leaf(f"{tag}_mat_{n}") for n in ("ambient_color", "diffuse_color", "specular_color", "shininess")}
    camt = leaf("camera_center")
    L = sh.Lights(lt["ambient_color"], lt["diffuse_color"], lt["specular_color"], **{vec: lt[vec]})
    M = sh.Materials(mt["ambient_color"], mt["diffuse_color"], mt["specular_color"], mt["shininess"])
    mesh = MeshView(g["verts"].to(d), g["faces"].to(d), g["normals"].to(d))
    col = p3d.phong_shading(mesh, Frag(g["pix_to_face"].to(d), g["bary"].to(d)), L, Cam(camt), M, g["texels"].to(d))
    pre = f"{tag}_pg_"
    assert torch.allclose(col.cpu(), g[pre + "colors"], atol=1e-5, rtol=1e-5)
    col.backward(g[pre + "grad_colors"].to(d))
    checks = [(camt.grad, "grad_camera")] + [(v.grad, "grad_light_" + k) for k, v in lt.items()] + \
             [(v.grad, "grad_mat_" + k) for k, v in mt.items()]
    for got, name in checks:
        ref = g[pre + name]
        assert _close(got.cpu().reshape(ref.shape), ref), (name, got, ref)


@pytest.mark.parametrize("tag", ["point", "dir"])
def test_flat_and_gouraud_mirrors_vs_reference_fixture(tag):
    import pytorch3d_amd as p3d

    g = U.shading_golden()
    d = torch.device("cuda:0")
    L, M = _lights_of(g, tag, d)
    cam = Cam(g["camera_center"].to(d))
    nv = [int(x) for x in g["num_verts_per_mesh"]]
    faces = g["faces"].to(d)
    off = 0
    faces_l = []
    for n_ in nv:  # back to per-mesh vertex indices
        sel = (g["faces"][:, 0] >= off) & (g["faces"][:, 0] < off + n_)
        faces_l.append((g["faces"][sel] - off).to(d))
        off += n_
    # flat: per-face lighting
    vl = [x.to(d).requires_grad_(True) for x in g["verts"].split(nv)]
    m = p3d.PackedMeshes(vl, faces_l)
    assert torch.equal(m.faces_packed(), faces)
    t_in = g["texels"].to(d).requires_grad_(True)
    b = g["bary"].to(d).requires_grad_(True)
    col = p3d.flat_shading(m, Frag(g["pix_to_face"].to(d), b), L, cam, M, t_in)
    ref = g[f"{tag}_flat_colors"]
    assert torch.allclose(col.cpu(), ref, atol=1e-5, rtol=1e-5), (col.cpu() - ref).abs().max()
    col.backward(g[f"{tag}_flat_grad_colors"].to(d))
    assert _close(torch.cat([x.grad for x in vl]).cpu(), g[f"{tag}_flat_grad_verts"])
    assert _close(t_in.grad.cpu(), g[f"{tag}_flat_grad_tex"])
    assert b.grad is None or b.grad.abs().max() == 0  # per-face values: no gradient to the barycentrics
    # Gouraud: per-vertex lighting, interpolated
    vl = [x.to(d).requires_grad_(True) for x in g["verts"].split(nv)]
    m = p3d.PackedMeshes(vl, faces_l)
    vc = g["verts_colors"].to(d).requires_grad_(True)
    b = g["bary"].to(d).requires_grad_(True)
    col = p3d.gouraud_shading(m, Frag(g["pix_to_face"].to(d), b), L, cam, M, verts_colors_packed=vc)
    ref = g[f"{tag}_gouraud_colors"]
    assert torch.allclose(col.cpu(), ref, atol=1e-5, rtol=1e-5), (col.cpu() - ref).abs().max()
    col.backward(g[f"{tag}_gouraud_grad_colors"].to(d))
    assert _close(torch.cat([x.grad for x in vl]).cpu(), g[f"{tag}_gouraud_grad_verts"])
    assert _close(vc.grad.cpu(), g[f"{tag}_gouraud_grad_tex"])
    assert _close(b.grad.cpu(), g[f"{tag}_gouraud_grad_bary"])


@pytest.mark.parametrize("K,size", [(1, (16, 16)), (3, (45, 37)), (8, (33, 64)), (10, (20, 50))])
@pytest.mark.parametrize("point", [True, False])
@pytest.mark.parametrize("D", [6, 9])
def test_phong_kernels_vs_oracle(K, size, point, D):
    from pytorch3d_amd.shading import _PhongShade

    d = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(K * 100 + D + int(point))
    N, (H, W), F = 3, size, 70
    p2f = torch.randint(-1, F, (N, H, W, K), generator=gen)
    p2f[0, :2] = -1
    # neighbouring pixels share faces, as real fragments do (exercises the same-face merging of the backward)
    p2f[1] = p2f[1, :1, :1].expand(H, W, K)
    bary = torch.rand(N, H, W, K, 3, generator=gen)
    bary = bary / bary.sum(-1, keepdim=True)
    fa = torch.randn(F, 3, D, generator=gen)
    fa[:, :, 2] += 3.0
    fa[5, :, 3:6] = 0.0  # a degenerate normal: the eps branch of F.normalize and of its gradient
    if D == 9:
        fa[:, :, 6:9] = torch.rand(F, 3, 3, generator=gen)
    texels = torch.rand(N, H, W, K, 3, generator=gen) if D == 6 else None
    params = torch.rand(N, 25, generator=gen)
    params[:, 9:12] = torch.randn(N, 3, generator=gen) * 2
    params[:, 22:25] = torch.randn(N, 3, generator=gen) * 2 - torch.tensor([0.0, 0.0, 4.0])
    params[:, 21] = torch.tensor([0.0, 1.0, 17.5])  # pow edge cases: exponent 0 and 1
    fa_g = fa.to(d).requires_grad_(True)
    b_g = bary.to(d).requires_grad_(True)
    t_g = texels.to(d).requires_grad_(True) if texels is not None else None
    col = _PhongShade.apply(p2f.to(d), b_g, fa_g, t_g, params.to(d), int(point))
    ref = orc.phong_shade(p2f, bary, fa, texels, params, point)
    assert torch.allclose(col.cpu(), ref, atol=1e-5, rtol=1e-5), (col.cpu() - ref).abs().max()
    go = torch.randn(N, H, W, K, 3, generator=gen)
    col.backward(go.to(d))
    rb, rf, rt, rp = orc.phong_shade_backward(go, p2f, bary, fa, texels, params, point, with_params=True)
    assert _close(b_g.grad.cpu(), rb)
    assert _close(fa_g.grad.cpu(), rf)
    if D == 6:
        assert _close(t_g.grad.cpu(), rt)
    # the same backward with the parameter gradient requested (sums over all samples of an image: scaled tolerance)
    p_g = params.to(d).requires_grad_(True)
    _PhongShade.apply(p2f.to(d), bary.to(d), fa.to(d), texels.to(d) if texels is not None else None, p_g,
                      int(point)).backward(go.to(d))
    assert torch.allclose(p_g.grad.cpu(), rp, rtol=2e-3, atol=2e-4 * rp.abs().max().item()), (p_g.grad.cpu() - rp).abs().max()


def test_phong_background_empty_and_errors():
    import pytorch3d_amd as p3d
    import pytorch3d_amd.shading as sh

    d = torch.device("cuda:0")
    v = torch.rand(10, 3, device=d)
    f = torch.randint(0, 10, (6, 3), device=d)
    mesh = MeshView(v, f, torch.nn.functional.normalize(torch.randn(10, 3, device=d), dim=1))
    cam = Cam(torch.tensor([[0.0, 0.0, -3.0]], device=d))
    L = sh.Lights(ambient_color=torch.tensor([[0.5, 0.5, 0.5]]), diffuse_color=torch.tensor([[0.3, 0.3, 0.3]]),
                  specular_color=torch.tensor([[0.2, 0.2, 0.2]]), location=torch.tensor([[0.0, 1.0, 0.0]]))
    M = sh.Materials(torch.ones(1, 3), torch.ones(1, 3), torch.ones(1, 3), torch.tensor([64.0]))
    p2f = torch.full((2, 5, 7, 2), -1, dtype=torch.int64, device=d)
    bary = torch.full((2, 5, 7, 2, 3), -1.0, device=d)
    tex = torch.rand(2, 5, 7, 2, 3, device=d)
    col = p3d.phong_shading(mesh, Frag(p2f, bary), L, cam, M, tex)
    # background: points = normals = 0 -> no diffuse, no specular: colour = ambient * texel (shading.py:96)
    assert torch.allclose(col, 0.5 * tex, atol=1e-7)
    col = p3d.phong_shading_vertex_colors(mesh, Frag(p2f, bary), L, cam, M, torch.rand(10, 3, device=d))
    assert (col == 0).all()
    e = p3d.phong_shading(mesh, Frag(p2f[:0], bary[:0]), L, cam, M, tex[:0])
    assert e.shape == (0, 5, 7, 2, 3)
    with pytest.raises(ValueError):
        p3d.phong_shading(mesh, Frag(p2f, bary), L._replace(ambient_color=torch.ones(3, 3)), cam, M, tex)
    with pytest.raises(RuntimeError):
        p3d.phong_shading(mesh, Frag(p2f.cpu(), bary.cpu()), L, cam, M, tex)


def _dense_phong(p2f, bary, fv, fn, texels, L_loc, la, ld, ls, ma, md, ms, shin, cam):
    """shading.py:59-96 + lighting.py:17-159 with torch ops, point light, (N,3) lights and (1,3) materials."""
    from pytorch3d_amd import interpolate_face_attributes as interp
    import torch.nn.functional as Fn

    pts = interp(p2f, bary, fv)
    nrm = interp(p2f, bary, fn)
    e = lambda t: t[:, None, None, None, :]
    direction = e(L_loc) - pts
    n_ = Fn.normalize(nrm, p=2, dim=-1, eps=1e-6)
    d_ = Fn.normalize(direction, p=2, dim=-1, eps=1e-6)
    cos = (n_ * d_).sum(-1)
    light_diffuse = e(ld) * torch.relu(cos)[..., None]
    mask = (cos > 0).float()
    view = Fn.normalize(e(cam) - pts, p=2, dim=-1, eps=1e-6)
    refl = -d_ + 2 * (cos[..., None] * n_)
    alpha = torch.relu((view * refl).sum(-1)) * mask
    light_spec = e(ls) * torch.pow(alpha, shin)[..., None]
    ambient = e(ma * la)
    return (ambient + md * light_diffuse) * texels + ms * light_spec


def test_phong_at_bench_fragment_size_vs_dense_torch():
    """N=4, 512x512, K=8 fragments of the bench generator."""
    import math

    import pytorch3d_amd as p3d
    import pytorch3d_amd.shading as sh

    d = torch.device("cuda:0")
    N = 4
    verts, faces = U.hetero_batch(N, seed=9)
    m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
    blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
    p2f, zbuf, bary, dists = p3d.rasterize_meshes(m, image_size=512, blur_radius=blur, faces_per_pixel=8,
                                                  perspective_correct=True, clip_barycentric_coords=True)
    gen = torch.Generator().manual_seed(3)
    faces_p = m.faces_packed()
    v0 = m.verts_packed().detach()
    n0 = m.verts_normals_packed().detach()
    texels0 = torch.rand(N, 512, 512, 8, 3, generator=gen).to(d)
    go = torch.randn(N, 512, 512, 8, 3, generator=gen).to(d)
    loc = (torch.randn(N, 3, generator=gen) * 2).to(d)
    la, ld, ls = (torch.rand(N, 3, generator=gen).to(d) for _ in range(3))
    ma, md, ms = (torch.rand(1, 3, generator=gen).to(d) for _ in range(3))
    shin = torch.tensor([12.0], device=d)
    cam = (torch.randn(N, 3, generator=gen) - torch.tensor([0.0, 0.0, 3.0])).to(d)

    res = []
    for fused in (True, False):
        v, nrm, t, b = (x.clone().requires_grad_(True) for x in (v0, n0, texels0, bary))
        if fused:
            col = p3d.phong_shading(MeshView(v, faces_p, nrm), Frag(p2f, b), sh.Lights(la, ld, ls, location=loc), Cam(cam),
                                    sh.Materials(ma, md, ms, shin), t)
        else:
            col = _dense_phong(p2f, b, v[faces_p], nrm[faces_p], t, loc, la, ld, ls, ma, md, ms, shin, cam)
        col.backward(go)
        res.append((col.detach(), v.grad, nrm.grad, t.grad, b.grad))
        del col
    (c1, gv1, gn1, gt1, gb1), (c2, gv2, gn2, gt2, gb2) = res
    assert torch.allclose(c1, c2, atol=2e-5, rtol=1e-4), (c1 - c2).abs().max()
    assert (c1[p2f < 0] == (ma * la)[:, None, None, None, :].expand_as(c1)[p2f < 0] * texels0[p2f < 0]).all()
    assert _close(gt1, gt2) and _close(gb1, gb2, atol=1e-4)
    # per-vertex sums of ~1e5 float terms in different orders
    assert _close(gv1, gv2, rtol=5e-3, atol=1e-3) and _close(gn1, gn2, rtol=5e-3, atol=1e-3)
