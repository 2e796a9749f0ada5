"""CPU suite (build container only: needs the read-only reference checkout): the two HOST-side replacements of
shim.install(patch_python=True) that carry no kernel -- the lean `Meshes.offset_verts` / `offset_verts_` and the cached camera
matrices of the patched `MeshRasterizer.forward` -- against the reference's own methods, on CPU tensors, in a subprocess (the
shim must not leak into the other tests).  Reference: pytorch3d/structures/meshes.py:1295-1360, renderer/mesh/rasterizer.py:171-216.
"""
import os
import subprocess
import sys
import textwrap

import pytest

REFERENCE = os.environ.get("P3D_REFERENCE_ROOT", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "pytorch3d", "renderer")),
                                reason="reference checkout not present (GPU box)")

SCRIPT = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    sys.path.insert(0, %r + "/tests")
    import torch
    import run_reference_suite as rrs
    rrs._stub_missing_packages()
    import pytorch3d_amd.shim as shim
    shim.install(%r)
    from pytorch3d.structures import Meshes
    from pytorch3d.renderer import FoVPerspectiveCameras, FoVOrthographicCameras, PerspectiveCameras, TexturesVertex, look_at_view_transform
    from pytorch3d.utils import ico_sphere, torus

    ref_offset, ref_offset_ = Meshes.offset_verts, Meshes.offset_verts_
    shim._patch_meshes_offset_verts()
    assert Meshes.offset_verts is not ref_offset

    def batch(padded=False):
        a, b, c = ico_sphere(1), torus(0.3, 1.0, 6, 9), ico_sphere(0)
        vl = a.verts_list() + b.verts_list() + c.verts_list()
        fl = a.faces_list() + b.faces_list() + c.faces_list()
        tex = TexturesVertex([torch.rand(v.shape[0], 3) for v in vl])
        m = Meshes(verts=vl, faces=fl, textures=tex)
        if padded:
            m = Meshes(verts=m.verts_padded(), faces=m.faces_padded())
        return m

    def same(x, y):
        assert x.shape == y.shape and torch.equal(x, y), (x.shape, y.shape)

    def compare(m_new, m_ref):
        same(m_new.verts_packed(), m_ref.verts_packed())
        same(m_new.verts_padded(), m_ref.verts_padded())
        same(m_new.faces_packed(), m_ref.faces_packed())
        same(m_new.faces_padded(), m_ref.faces_padded())
        for x, y in zip(m_new.verts_list(), m_ref.verts_list()):
            same(x, y)
        for x, y in zip(m_new.faces_list(), m_ref.faces_list()):
            same(x, y)
        same(m_new.num_verts_per_mesh(), m_ref.num_verts_per_mesh())
        same(m_new.mesh_to_faces_packed_first_idx(), m_ref.mesh_to_faces_packed_first_idx())
        same(m_new.edges_packed(), m_ref.edges_packed())
        assert torch.allclose(m_new.verts_normals_packed(), m_ref.verts_normals_packed(), atol=1e-6)
        assert torch.allclose(m_new.faces_normals_packed(), m_ref.faces_normals_packed(), atol=1e-6)
        assert torch.allclose(m_new.faces_areas_packed(), m_ref.faces_areas_packed(), atol=1e-7)
        assert len(m_new) == len(m_ref) and m_new._V == m_ref._V and m_new._F == m_ref._F and m_new.equisized == m_ref.equisized

    gen = torch.Generator().manual_seed(0)
    for padded in (False, True):
        for warm in (False, True):  # with and without the normals / areas caches on the source
            m = batch(padded)
            if warm:
                m.verts_normals_packed(), m.faces_areas_packed(), m.verts_padded()
            V = m.verts_packed().shape[0]
            off = torch.randn(V, 3, generator=gen) * 0.1
            before = m.verts_packed().clone()
            new, ref = m.offset_verts(off), ref_offset(m, off)
            compare(new, ref)
            same(m.verts_packed(), before)  # out of place
            if not padded:
                same(new.textures.verts_features_packed(), ref.textures.verts_features_packed())
                assert new.textures is not m.textures
            # a second offset on the result, the (3,) form, the in-place form
            off3 = torch.tensor([0.1, -0.2, 0.3])
            compare(new.offset_verts(off3), ref_offset(ref, off3))
            m2, r2 = batch(padded), batch(padded)
            if warm:
                m2.verts_normals_packed(), r2.verts_normals_packed()
            assert m2.offset_verts_(off) is m2
            ref_offset_(r2, off)
            compare(m2, r2)
    # gradients reach the offsets and the source vertices
    m = batch()
    off = torch.zeros(m.verts_packed().shape[0], 3, requires_grad=True)
    m.offset_verts(off).verts_packed().square().sum().backward()
    assert torch.allclose(off.grad, 2 * m.verts_packed())
    assert shim.PATCH_CALLS["Meshes.offset_verts"][0] >= 9 and shim.PATCH_CALLS["Meshes.offset_verts"][1] == 0
    # refused forms go to the reference's own method (which raises its own error)
    try:
        m.offset_verts(torch.zeros(5, 3))
    except ValueError as e:
        assert "Verts offsets must have dimension" in str(e)
    else:
        raise AssertionError("shape mismatch was accepted")
    assert shim.PATCH_CALLS["Meshes.offset_verts"][1] == 1
    m.offset_verts(torch.zeros(m.verts_packed().shape[0], 3, dtype=torch.float64))  # dtype: the reference's path
    assert shim.PATCH_CALLS["Meshes.offset_verts"][1] == 2

    # ---- camera matrices: cached == fresh, and every way of changing the camera invalidates ----
    def fresh(c):
        w2v = c.get_world_to_view_transform().get_matrix()
        v2n = c.get_projection_transform().compose(c.get_ndc_camera_transform()).get_matrix()
        return w2v, v2n

    R, T = look_at_view_transform(dist=[2.7, 3.0], elev=[10, 20], azim=[30, -40])
    for cams in (FoVPerspectiveCameras(R=R, T=T, znear=0.5), FoVOrthographicCameras(R=R, T=T),
                 PerspectiveCameras(R=R, T=T, focal_length=((1.2, 1.3), (0.9, 1.0)), principal_point=((0.1, 0.0), (0.0, -0.1)))):
        a = shim.camera_matrices(cams, {})
        b = shim.camera_matrices(cams, {"cameras": cams})
        assert a is b, "the second call did not come from the cache"
        w2v, v2n = fresh(cams)
        assert torch.equal(a[0], w2v) and torch.equal(a[1], v2n) and a[2] == cams.is_perspective()
        # override in the call: never cached
        c = shim.camera_matrices(cams, {"T": T + 1.0})
        assert c is not a and not torch.equal(c[0], a[0])
        # (the reference's get_world_to_view_transform STORES an overriding R / T on the camera, cameras.py: `self.T = T`:
        # the camera has changed, and the next plain call must see that)
        c2 = shim.camera_matrices(cams, {})
        assert c2 is not a and torch.equal(c2[0], fresh(cams)[0]) and torch.equal(c2[0], c[0])
        assert shim.camera_matrices(cams, {}) is c2
        cams.T = T + 0.5  # a new tensor
        d = shim.camera_matrices(cams, {})
        assert d is not c2 and torch.equal(d[0], fresh(cams)[0]) and not torch.equal(d[0], c2[0])
        cams.R.mul_(-1.0)  # an in-place edit
        e = shim.camera_matrices(cams, {})
        assert e is not d and torch.equal(e[0], fresh(cams)[0]) and not torch.equal(e[0], d[0])
        if hasattr(cams, "znear") and torch.is_tensor(cams.znear):
            cams.znear = cams.znear * 2
            f = shim.camera_matrices(cams, {})
            assert f is not e and abs(f[3] - float(cams.znear.min())) < 1e-7 and torch.equal(f[1], fresh(cams)[1])
    # cameras under optimisation are not cached (and the patched forward leaves them to the reference's autograd path)
    Tg = T.clone().requires_grad_(True)
    cg = FoVPerspectiveCameras(R=R, T=Tg)
    g1 = shim.camera_matrices(cg, {})
    assert g1[0].requires_grad and shim.camera_matrices(cg, {}) is not g1
    # ADVICE round 4: tensors a camera keeps in the nn.Module registries are part of the fingerprint, Parameters switch the cache off
    cb = FoVPerspectiveCameras(R=R, T=T)
    del cb.T
    cb.register_buffer("T", T.clone())
    b1 = shim.camera_matrices(cb, {})
    assert shim.camera_matrices(cb, {}) is b1
    cb.T.add_(0.25)  # in-place edit of a BUFFER
    b2 = shim.camera_matrices(cb, {})
    assert b2 is not b1 and torch.equal(b2[0], fresh(cb)[0]) and not torch.equal(b2[0], b1[0])
    cp = FoVPerspectiveCameras(R=R, T=T)
    del cp.T
    cp.T = torch.nn.Parameter(T.clone())
    p1 = shim.camera_matrices(cp, {})
    with torch.no_grad():
        cp.T.data.add_(0.5)  # bypasses the version counter: only safe because Parameters are never cached
    p2 = shim.camera_matrices(cp, {})
    assert p2 is not p1 and torch.equal(p2[0].detach(), fresh(cp)[0].detach()) and not torch.equal(p2[0].detach(), p1[0].detach())
    # ADVICE round 4: the lean offset_verts is for Meshes itself (a subclass goes through the reference's clone()), and the copy's
    # lists are its own
    class MyMeshes(Meshes):
        pass
    sub = MyMeshes(verts=batch().verts_list(), faces=batch().faces_list())
    before = shim.PATCH_CALLS["Meshes.offset_verts"][1]
    sub.offset_verts(torch.zeros(sub.verts_packed().shape[0], 3))
    assert shim.PATCH_CALLS["Meshes.offset_verts"][1] == before + 1
    m0 = batch()
    m1 = m0.offset_verts(torch.zeros(m0.verts_packed().shape[0], 3))
    assert m1.faces_list() is not m0.faces_list() and m1._faces_list is not m0._faces_list
    m1.faces_list().append(torch.zeros((1, 3), dtype=torch.int64))
    assert len(m0.faces_list()) == 3
    # ---- hard shaders on the nearest slot only: same image, same gradients as the reference's all-K evaluation ----
    from pytorch3d.renderer import HardPhongShader, HardGouraudShader, HardFlatShader, PointLights
    from pytorch3d.renderer.mesh.rasterizer import Fragments
    import pytorch3d.renderer.mesh.shader as shader_mod
    originals = {n: getattr(shader_mod, n).forward for n in ("HardPhongShader", "HardGouraudShader", "HardFlatShader")}
    shim._patch_hard_and_silhouette_shaders()
    m = batch()
    Fn = m.faces_packed().shape[0]
    g2 = torch.Generator().manual_seed(5)
    N, H, W, K = len(m), 9, 11, 4
    p2f = torch.randint(-1, Fn, (N, H, W, K), generator=g2)
    p2f[:, :2] = -1  # background rows
    bary = torch.rand(N, H, W, K, 3, generator=g2)
    bary = bary / bary.sum(-1, keepdim=True)
    zbuf, dists = torch.rand(N, H, W, K, generator=g2) + 1.0, torch.rand(N, H, W, K, generator=g2) * 1e-3
    cams1 = FoVPerspectiveCameras(R=R[:1], T=T[:1])
    lights = PointLights(location=[[0.0, 1.0, -2.0]])
    gout = torch.randn(N, H, W, 4, generator=g2)
    for name, cls in (("HardPhongShader", HardPhongShader), ("HardGouraudShader", HardGouraudShader), ("HardFlatShader", HardFlatShader)):
        res = []
        for fwd in (cls.forward, originals[name]):
            b = bary.clone().requires_grad_(True)
            vcol = m.textures.verts_features_packed().clone().requires_grad_(True)
            mm = Meshes(verts=m.verts_list(), faces=m.faces_list(), textures=TexturesVertex(list(vcol.split(m.num_verts_per_mesh().tolist()))))
            sh = cls(cameras=cams1, lights=lights)
            img = fwd(sh, Fragments(pix_to_face=p2f, zbuf=zbuf, bary_coords=b, dists=dists), mm)
            (img * gout).sum().backward()
            res.append((img.detach(), b.grad.clone(), vcol.grad.clone()))
        for x, y in zip(res[0], res[1]):
            assert torch.equal(x, y) or torch.allclose(x, y, atol=1e-6, rtol=1e-5), (name, float((x - y).abs().max()))
        assert float(res[0][1][..., 1:, :].abs().max()) == 0.0  # slots behind the nearest get no gradient, as in the reference
        assert shim.PATCH_CALLS[name + ".forward"][0] >= 1
    shim.uninstall_python_patches()
    assert Meshes.offset_verts is ref_offset and Meshes.offset_verts_ is ref_offset_
    assert all(getattr(shader_mod, n).forward is f for n, f in originals.items())
    print("HOST-PATCHES-OK")
""")


def test_lean_offset_verts_and_cached_camera_matrices_equal_the_reference():
    out = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, ROOT, REFERENCE)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "HOST-PATCHES-OK" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])
