"""SURVEY.md 8(f) row 3, second half: the world -> NDC camera transform fused into the face gather.

Fixture: tests/golden/cow_ref.npz -- the reference's cow, its FoVPerspectiveCameras matrices and the NDC vertices its own
`MeshRasterizer.transform` produced (renderer/mesh/rasterizer.py:171-216), plus its rasterization of them."""
import os

import numpy as np
import pytest
import torch

import _util as U

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _cow(d):
    g = np.load(os.path.join(U.GOLDEN, "cow_ref.npz"))
    t = lambda k: torch.from_numpy(g[k])
    return g, t("verts_world").to(d), t("faces").long().to(d), t("world_to_view").to(d), t("projection").to(d), t("verts_ndc").to(d)


def test_transform_kernels_reproduce_the_reference_transform():
    import ctypes

    from pytorch3d_amd import _C, _lib

    d = _dev()
    g, vw, faces, w2v, proj, ndc_ref = _cow(d)
    lib = _lib.load()
    V, F = vw.shape[0], faces.shape[0]
    mats = torch.stack([w2v, proj], 0)[None].contiguous()
    first = torch.zeros(1, dtype=torch.int64, device=d)
    out = torch.empty((V, 3), device=d)
    _lib.check(lib.p3d_transform_verts_forward(_C._ptr(vw), _C._ptr(first), _C._ptr(mats), V, 1, 1, _C._ptr(out), _C._stream(d)), "fwd")
    err = (out - ndc_ref).abs().max().item()
    print(f"[transform] max |verts_ndc - MeshRasterizer.transform| = {err:.2e}")
    assert err <= 1e-5
    fv = torch.empty((F, 3, 3), device=d)
    _lib.check(lib.p3d_transform_gather_face_verts(_C._ptr(vw), _C._ptr(faces), _C._ptr(first), _C._ptr(mats), V, F, 1, 1,
                                                   _C._ptr(fv), _C._stream(d)), "gather")
    assert torch.equal(fv, out[faces])  # same arithmetic per corner as per vertex


def test_rasterize_meshes_world_matches_reference_pipeline_and_torch_autograd():
    import pytorch3d_amd as p3d
    from pytorch3d_amd.rasterize_meshes import transform_points_reference

    d = _dev()
    g, vw, faces, w2v, proj, ndc_ref = _cow(d)
    H, K, blur = int(g["image_size"]), int(g["K"]), float(g["blur_radius"])
    vw = vw.clone().requires_grad_(True)
    m = p3d.PackedMeshes([vw], [faces])
    out = p3d.rasterize_meshes_world(m, w2v, proj, image_size=H, blur_radius=blur, faces_per_pixel=K, perspective_correct=True,
                                     clip_barycentric_coords=True)
    ref_idx = torch.from_numpy(g["pix_to_face"]).long().to(d)
    same = out[0] == ref_idx
    print(f"[world raster] pix_to_face differences vs the reference pipeline: {int((~same).sum())} / {same.numel()}")
    assert (~same).float().mean().item() < 2e-4  # NDC vertices differ by an ulp or two from torch's bmm
    assert torch.allclose(out[1][same], torch.from_numpy(g["zbuf"]).to(d)[same], atol=1e-4)
    # gradient to the world vertices vs torch autograd through the torch formulation of the transform + our NDC path
    gen = torch.Generator().manual_seed(5)
    gz = torch.randn(out[1].shape, generator=gen).to(d)
    gb = torch.randn(out[2].shape, generator=gen).to(d)
    gd = torch.randn(out[3].shape, generator=gen).to(d)
    (g_fused,) = torch.autograd.grad([out[1], out[2], out[3]], vw, [gz, gb, gd])
    vw2 = vw.detach().clone().requires_grad_(True)
    ndc = transform_points_reference(vw2, torch.zeros(vw2.shape[0], dtype=torch.int64, device=d), w2v[None], proj[None])
    o2 = p3d.rasterize_meshes(p3d.PackedMeshes([ndc], [faces]), image_size=H, blur_radius=blur, faces_per_pixel=K,
                              perspective_correct=True, clip_barycentric_coords=True)
    (g_ref,) = torch.autograd.grad([o2[1], o2[2], o2[3]], vw2, [gz, gb, gd])
    # the cow has slivers with huge, ill-conditioned gradients: compare vertices whose gradient is of ordinary size
    okv = (g_ref.abs().max(1).values < 1e3) & (g_fused.abs().max(1).values < 1e3)
    scale = float(g_ref[okv].abs().max())
    bad = int((~torch.isclose(g_fused[okv], g_ref[okv], rtol=5e-3, atol=5e-4 * scale)).sum())
    print(f"[world raster backward] {int(okv.sum())} / {okv.numel()} vertices of ordinary gradient size, beyond rtol 5e-3: {bad}")
    assert bad <= 3e-3 * okv.sum().item() * 3
    # batch of two different cameras + a camera that requires grad (torch path)
    w2v2 = w2v.clone()
    w2v2[3, 2] += 0.3
    m2 = p3d.PackedMeshes([vw.detach(), vw.detach() * 0.9], [faces, faces])
    a = p3d.rasterize_meshes_world(m2, torch.stack([w2v, w2v2]), proj, image_size=64, faces_per_pixel=2)
    cam = torch.stack([w2v, w2v2]).requires_grad_(True)
    b = p3d.rasterize_meshes_world(m2, cam, proj, image_size=64, faces_per_pixel=2)
    assert (a[0] != b[0]).float().mean().item() < 1e-3
    b[1].sum().backward()
    assert cam.grad is not None and torch.isfinite(cam.grad).all()
