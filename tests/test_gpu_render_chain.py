"""End-to-end parity of the chain this repository accelerates -- rasterize_meshes (+ the no-op near-plane clip the
rasterizer applies for perspective cameras) -> phong_shading with TexturesVertex colours -> softmax_rgb_blend --
against the image and the colour gradient of the REFERENCE's own MeshRenderer(MeshRasterizer, SoftPhongShader) on CPU
(tests/golden/make_golden_render.py; camera transforms are out of scope and come with the fixture as NDC vertices).
"""
import os
from collections import namedtuple

import numpy as np
import pytest
import torch

import _util as U

pytestmark = pytest.mark.gpu
Frag = namedtuple("Frag", "pix_to_face zbuf bary_coords dists")


class Cam:
    def __init__(self, c):
        self.c = c

    def get_camera_center(self):
        return self.c


def test_soft_phong_render_matches_reference_renderer():
    import pytorch3d_amd as p3d
    import pytorch3d_amd.shading as sh

    g = np.load(os.path.join(U.GOLDEN, "render_ref.npz"))
    t = lambda k: torch.from_numpy(g[k])
    d = torch.device("cuda:0")
    nv, nf = [int(x) for x in g["num_verts"]], [int(x) for x in g["num_faces"]]
    faces_l, off = [], 0
    for f, n in zip(t("faces").split(nf), nv):
        faces_l.append((f - off).to(d))
        off += n
    ndc = p3d.PackedMeshes([v.to(d) for v in t("verts_ndc").split(nv)], faces_l)
    H, K = int(g["image_size"]), int(g["K"])
    frag = Frag(*p3d.rasterize_meshes(ndc, image_size=H, blur_radius=float(g["blur_radius"]), faces_per_pixel=K,
                                      perspective_correct=True, clip_barycentric_coords=True, cull_backfaces=False,
                                      z_clip_value=float(g["znear"]) / 2))  # rasterizer.py:244-251
    # the fragments themselves: face ids as the reference's CPU rasterizer found them (ulp-level depth ties aside)
    same = (frag.pix_to_face.cpu() == t("pix_to_face")).float().mean().item()
    assert same > 0.999, same
    world = p3d.PackedMeshes([v.to(d) for v in t("verts_world").split(nv)], faces_l)
    vcol = t("verts_colors").to(d).requires_grad_(True)
    L = sh.Lights(t("light_ambient").to(d), t("light_diffuse").to(d), t("light_specular").to(d),
                  location=t("light_location").to(d))
    M = sh.Materials(torch.ones(1, 3, device=d), torch.ones(1, 3, device=d), torch.ones(1, 3, device=d),
                     t("shininess").to(d))
    colors = p3d.phong_shading_vertex_colors(world, frag, L, Cam(t("camera_center").to(d)), M, vcol)
    bp = p3d.BlendParams(float(g["sigma"]), float(g["gamma"]), tuple(float(x) for x in g["background"]))
    zn = torch.full((2,), float(g["znear"]), device=d)
    zf = torch.full((2,), float(g["zfar"]), device=d)
    img = p3d.softmax_rgb_blend(colors, frag, bp, znear=zn, zfar=zf)
    ref = t("image")
    err = (img.cpu() - ref).abs()
    # alpha exactly as tight as the stage tests; RGB of the few pixels whose face lists differ by a tie may move
    assert err[..., 3].max() < 1e-5
    assert (err[..., :3] > 2e-4).float().mean() < 2e-3, err[..., :3].max()
    img.backward(t("grad_image").to(d))
    rg = t("grad_verts_colors")
    assert torch.allclose(vcol.grad.cpu(), rg, rtol=2e-3, atol=2e-4 * rg.abs().max().item())


@pytest.mark.parametrize("tag", ["alpha", "norm"])
def test_points_render_matches_reference_renderer(tag):
    """rasterize_points -> (1 - d2 / r2) weights -> compositor, against the reference's PointsRenderer(PointsRasterizer,
    AlphaCompositor | NormWeightedCompositor) on CPU (tests/golden/make_golden_render_points.py)."""
    import pytorch3d_amd as p3d

    g = np.load(os.path.join(U.GOLDEN, "render_points_ref.npz"))
    t = lambda k: torch.from_numpy(g[k])
    d = torch.device("cuda:0")
    npts = [int(x) for x in g["num_points"]]
    pc = p3d.PackedPointclouds([p.to(d) for p in t("points_ndc").split(npts)])
    H, W = (int(x) for x in g["image_size"])
    r, K = float(g["radius"]), int(g["K"])
    idx, zbuf, dists = p3d.rasterize_points(pc, image_size=(H, W), radius=r, points_per_pixel=K)
    assert (idx.cpu() == t("idx")).float().mean().item() > 0.999
    assert torch.allclose(zbuf.cpu(), t("zbuf"), atol=1e-5) or (zbuf.cpu() - t("zbuf")).abs().gt(1e-5).float().mean() < 1e-3
    feats = t("features").to(d).requires_grad_(True)
    weights = 1 - dists.permute(0, 3, 1, 2) / (r * r)  # points/renderer.py:64-65
    fn = p3d.alpha_composite if tag == "alpha" else p3d.norm_weighted_sum
    img = fn(idx.long().permute(0, 3, 1, 2), weights, feats.permute(1, 0))
    if tag == "alpha":  # compositor.py:66-110 (_add_background_color_to_images), plain torch glue
        bg = torch.tensor([0.1, 0.2, 0.3, 1.0], device=d)
        img = torch.where((idx[..., 0] < 0)[:, None], bg[None, :, None, None], img)
    img = img.permute(0, 2, 3, 1)
    ref = t(f"{tag}_image")
    assert ((img.cpu() - ref).abs() > 1e-5).float().mean() < 2e-3, (img.cpu() - ref).abs().max()
    img.backward(t(f"{tag}_grad_image").to(d))
    rg = t(f"{tag}_grad_features")
    assert torch.allclose(feats.grad.cpu(), rg, rtol=2e-3, atol=2e-4 * rg.abs().max().item())


@pytest.mark.parametrize("bin_size", [0, None])
@pytest.mark.parametrize("tag", ["alpha", "norm"])
def test_fused_points_render_matches_reference_renderer(tag, bin_size):
    """The same fixture through the fused node of round 6 (pytorch3d_amd.render_points: the compositor in the fine kernel's epilogue /
    as a pass behind the naive launch, one backward kernel): image and feature gradient against the reference's PointsRenderer on CPU,
    and bit-equal to the operator chain above."""
    import pytorch3d_amd as p3d

    g = np.load(os.path.join(U.GOLDEN, "render_points_ref.npz"))
    t = lambda k: torch.from_numpy(g[k])
    d = torch.device("cuda:0")
    npts = [int(x) for x in g["num_points"]]
    pc = p3d.PackedPointclouds([p.to(d) for p in t("points_ndc").split(npts)])
    H, W = (int(x) for x in g["image_size"])
    r, K = float(g["radius"]), int(g["K"])
    feats = t("features").to(d).requires_grad_(True)
    img, idx, zbuf, dists = p3d.render_points_alpha(pc, feats, image_size=(H, W), radius=r, points_per_pixel=K, bin_size=bin_size,
                                                    compositor=tag)
    assert (idx.cpu() == t("idx")).float().mean().item() > 0.999
    chain = (p3d.alpha_composite if tag == "alpha" else p3d.norm_weighted_sum)(
        idx.long().permute(0, 3, 1, 2), 1 - dists.permute(0, 3, 1, 2) / (r * r), feats.detach().permute(1, 0)).permute(0, 2, 3, 1)
    assert torch.equal(img.detach(), chain)
    out = img
    if tag == "alpha":  # compositor.py:66-110 (_add_background_color_to_images), plain torch glue
        bg = torch.tensor([0.1, 0.2, 0.3, 1.0], device=d)
        out = torch.where((idx[..., 0] < 0)[..., None], bg[None, None, None, :], img)
    ref = t(f"{tag}_image")
    assert ((out.detach().cpu() - ref).abs() > 1e-5).float().mean() < 2e-3, (out.detach().cpu() - ref).abs().max()
    out.backward(t(f"{tag}_grad_image").to(d))
    rg = t(f"{tag}_grad_features")
    assert torch.allclose(feats.grad.cpu(), rg, rtol=2e-3, atol=2e-4 * rg.abs().max().item())
