"""The pybind flavour of the drop-in boundary (pytorch3d_amd/csrc/bind.cpp: INTEGRATION.md section B compiled) against the
ctypes flavour (pytorch3d_amd/_C.py) -- the same 20 operators over the same C ABI: same bits (forward operators; gradients up to the
order of their float atomics), same error texts; and what a call costs on the host through either (BASELINE configs[1])."""
import os
import time

import numpy as np
import pytest
import torch

import _util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def both():
    from pytorch3d_amd import _C, build_bind

    try:
        pyb = build_bind.load()
    except Exception as e:  # noqa: BLE001
        pytest.skip("pybind flavour not built on this box: %r" % (e,))
    return _C, pyb


def _same(a, b, what):
    a = a if isinstance(a, (tuple, list)) else (a,)
    b = b if isinstance(b, (tuple, list)) else (b,)
    assert len(a) == len(b), what
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.dtype == y.dtype and x.shape == y.shape, (what, i, x.dtype, y.dtype, x.shape, y.shape)
        assert torch.equal(x.view(torch.uint8), y.view(torch.uint8)) if x.is_contiguous() and y.is_contiguous() else torch.equal(x, y), (what, i)


def _close(a, b, what, rtol=5e-3):
    a = a if isinstance(a, (tuple, list)) else (a,)
    b = b if isinstance(b, (tuple, list)) else (b,)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.shape == y.shape, (what, i)
        scale = max(float(y.abs().max()), 1e-12)
        assert float((x - y).abs().max()) <= rtol * scale, (what, i, float((x - y).abs().max()), scale)


def test_every_operator_returns_what_the_ctypes_flavour_returns(both):
    _C, pyb = both
    assert set(_C.HOT_PATH_EXPORTS) <= set(dir(pyb)) and pyb.__p3d_amd_flavour__ == "pybind"
    d = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    # ---- meshes: naive, fused, coarse, fine, backward ------------------------------------------------------------------
    F, N, K, size = 260, 2, 8, (72, 56)
    fv = U.triangle_soup(F, gen, size=0.6).to(d)
    first, count = [t.to(d) for t in U.split_counts(F, N)]
    nbr = torch.full((F,), -1, dtype=torch.int64, device=d)
    for a in range(0, F - 1, 16):
        nbr[a], nbr[a + 1] = a + 1, a
    for bin_size, M in ((0, 0), (16, 300), (32, 400)):
        args = (fv, first, count, nbr, size, 0.004, K, bin_size, M, True, True, False)
        _same(pyb.rasterize_meshes(*args), _C.rasterize_meshes(*args), f"rasterize_meshes bin {bin_size}")
    nargs = (fv, first, count, nbr, size, 0.004, 5, False, False, True)
    _same(pyb._rasterize_meshes_naive(*nargs), _C._rasterize_meshes_naive(*nargs), "_rasterize_meshes_naive")
    cargs = (fv, first, count, size, 0.004, 16, 300)
    bins_p, bins_c = pyb._rasterize_meshes_coarse(*cargs), _C._rasterize_meshes_coarse(*cargs)
    _same(bins_p, bins_c, "_rasterize_meshes_coarse")
    fargs = (fv, bins_c, nbr, size, 0.004, 16, K, True, True, False)
    out_p, out_c = pyb._rasterize_meshes_fine(*fargs), _C._rasterize_meshes_fine(*fargs)
    _same(out_p, out_c, "_rasterize_meshes_fine")
    g = [torch.randn(o.shape, generator=gen).to(d) for o in out_c[1:]]
    _close(pyb.rasterize_meshes_backward(fv, out_c[0], g[0], g[1], g[2], True, True),
           _C.rasterize_meshes_backward(fv, out_c[0].clone(), g[0], g[1], g[2], True, True), "rasterize_meshes_backward")
    # ---- points -------------------------------------------------------------------------------------------------------------
    P, Kp = 3000, 10
    pts = torch.cat([torch.rand(P, 2, generator=gen) * 2 - 1, torch.rand(P, 1, generator=gen) * 2 + 0.5], 1).to(d)
    pfirst, pcount = [t.to(d) for t in U.split_counts(P, 2)]
    rad = (torch.rand(P, generator=gen) * 0.04 + 0.01).to(d)
    for bin_size, M in ((0, 0), (16, 500)):
        pa = (pts, pfirst, pcount, size, rad, Kp, bin_size, M)
        _same(pyb.rasterize_points(*pa), _C.rasterize_points(*pa), f"rasterize_points bin {bin_size}")
    _same(pyb._rasterize_points_naive(pts, pfirst, pcount, size, rad, 7), _C._rasterize_points_naive(pts, pfirst, pcount, size, rad, 7),
          "_rasterize_points_naive")
    pb_p, pb_c = pyb._rasterize_points_coarse(pts, pfirst, pcount, size, rad, 16, 500), _C._rasterize_points_coarse(pts, pfirst, pcount, size, rad, 16, 500)
    _same(pb_p, pb_c, "_rasterize_points_coarse")
    po_p, po_c = pyb._rasterize_points_fine(pts, pb_c, size, rad, 16, Kp), _C._rasterize_points_fine(pts, pb_c, size, rad, 16, Kp)
    _same(po_p, po_c, "_rasterize_points_fine")
    gz, gd = (torch.randn(po_c[1].shape, generator=gen).to(d) for _ in range(2))
    _close(pyb.rasterize_points_backward(pts, po_c[0], gz, gd), _C.rasterize_points_backward(pts, po_c[0], gz, gd), "rasterize_points_backward")
    # ---- compositors: the renderer's permuted views and its transposed features, and contiguous tensors ------------------------
    C = 3
    idx, zbuf, dists = po_c
    alphas = (1 - dists / (rad.max() ** 2)).clamp(0, 1).permute(0, 3, 1, 2)
    pidx = idx.long().permute(0, 3, 1, 2)
    feats_pc = torch.rand(P, C, generator=gen).to(d)
    for feats in (feats_pc.t(), feats_pc.t().contiguous()):
        for al, pi in ((alphas, pidx), (alphas.contiguous(), pidx.contiguous())):
            for name in ("accum_alphacomposite", "accum_weightedsumnorm", "accum_weightedsum"):
                img_p, img_c = getattr(pyb, name)(feats, al, pi), getattr(_C, name)(feats, al, pi)
                _same(img_p, img_c, name)
                go = torch.randn(img_c.shape, generator=gen).to(d)
                bp, bc = getattr(pyb, name + "_backward")(go, feats, al, pi), getattr(_C, name + "_backward")(go, feats, al, pi)
                assert bp[0].stride() == bc[0].stride(), name  # grad_features comes back in the features' layout either way
                _close(bp, bc, name + "_backward")
    # ---- interpolate_face_attributes, f32 and f64 ----------------------------------------------------------------------------------
    p2f = out_c[0].reshape(-1)
    for dt in (torch.float32, torch.float64):
        bary = out_c[2].reshape(-1, 3).to(dt)
        attrs = torch.rand(F, 3, 5, generator=gen, dtype=torch.float64).to(dt).to(d)
        ia_p, ia_c = pyb.interp_face_attrs_forward(p2f, bary, attrs), _C.interp_face_attrs_forward(p2f, bary, attrs)
        _same(ia_p, ia_c, "interp_face_attrs_forward")
        gp = torch.randn(ia_c.shape, generator=gen, dtype=torch.float64).to(dt).to(d)
        _close(pyb.interp_face_attrs_backward(p2f, bary, attrs, gp), _C.interp_face_attrs_backward(p2f, bary, attrs, gp), "interp_face_attrs_backward")
    # ---- sigmoid alpha blend -------------------------------------------------------------------------------------------------------
    al_p, al_c = pyb.sigmoid_alpha_blend(out_c[3], out_c[0], 1e-3), _C.sigmoid_alpha_blend(out_c[3], out_c[0], 1e-3)
    _same(al_p, al_c, "sigmoid_alpha_blend")
    ga = torch.randn(al_c.shape, generator=gen).to(d)
    _same(pyb.sigmoid_alpha_blend_backward(ga, al_c, out_c[3], out_c[0], 1e-3), _C.sigmoid_alpha_blend_backward(ga, al_c, out_c[3], out_c[0], 1e-3),
          "sigmoid_alpha_blend_backward")


def test_error_texts_and_refusals_are_the_same(both):
    _C, pyb = both
    d = torch.device("cuda:0")
    fv = torch.rand(4, 3, 3, device=d)
    z = torch.zeros(1, dtype=torch.int64, device=d)
    nbr = torch.full((4,), -1, dtype=torch.int64, device=d)
    for mod in (_C, pyb):
        with pytest.raises(RuntimeError, match="Must have points_per_pixel <= 150"):
            mod.rasterize_meshes(fv, z, z + 4, nbr, (8, 8), 0.0, 151, 0, 0, False, False, False)
        with pytest.raises(RuntimeError, match="too many"):
            mod.rasterize_meshes(fv, z, z + 4, nbr, (64, 64), 0.0, 2, 2, 10, False, False, False)
        with pytest.raises(RuntimeError, match="face_verts must have dimensions"):
            mod.rasterize_meshes(fv[:, :2], z, z + 4, nbr, (8, 8), 0.0, 2, 0, 0, False, False, False)
        with pytest.raises(RuntimeError):  # a CPU tensor: refused, never emulated
            mod.rasterize_meshes(fv.cpu(), z.cpu(), z.cpu() + 4, nbr.cpu(), (8, 8), 0.0, 2, 0, 0, False, False, False)


def test_the_reference_rasterizer_runs_over_the_pybind_flavour(both):
    """`shim.install(flavour="pybind")`: the unmodified reference MeshRasterizer (when its Python package is staged on this box) over the
    compiled module -- same fragments as over the ctypes one."""
    import subprocess
    import sys

    root = U.ROOT
    stage = os.path.join(root, "oracle", "_ref", "reference_py")
    if not os.path.isdir(os.path.join(stage, "pytorch3d", "renderer")):
        pytest.skip("the reference's Python package is not staged on this box")
    code = r'''
import sys, json, hashlib, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import pytorch3d_amd.shim as shim
shim.install(%r, flavour=sys.argv[1])
import _util as U
from pytorch3d.structures import Meshes
from pytorch3d.renderer import MeshRasterizer, RasterizationSettings, FoVPerspectiveCameras, look_at_view_transform
d = torch.device("cuda:0")
v, f = U.ico_sphere(3)
R, T = look_at_view_transform(2.7, 10, 20)
cams = FoVPerspectiveCameras(device=d, R=R, T=T)
rs = RasterizationSettings(image_size=96, blur_radius=1e-3, faces_per_pixel=6, bin_size=None, perspective_correct=True, clip_barycentric_coords=True)
vv = v.to(d).requires_grad_(True)
fr = MeshRasterizer(cameras=cams, raster_settings=rs)(Meshes(verts=[vv], faces=[f.to(d)]))
(fr.zbuf[fr.pix_to_face >= 0].sum() + fr.dists[fr.pix_to_face >= 0].sum()).backward()
h = hashlib.sha256()
for t in (fr.pix_to_face, fr.zbuf, fr.bary_coords, fr.dists):
    h.update(t.detach().cpu().contiguous().numpy().tobytes())
print(json.dumps({"flavour": sys.modules["pytorch3d._C"].__p3d_amd_flavour__, "sha": h.hexdigest(), "gnorm": float(torch.nan_to_num(vv.grad, nan=0.0, posinf=0.0, neginf=0.0).double().abs().median())}))
''' % (root, root, stage)
    res = {}
    for fl in ("ctypes", "pybind"):
        r = subprocess.run([sys.executable, "-c", code, fl], capture_output=True, text=True, timeout=300, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        import json

        res[fl] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert res[fl]["flavour"] == fl
    assert res["ctypes"]["sha"] == res["pybind"]["sha"]
    # (the median |gradient|: edge-on faces of the sphere reach 1e30 and more, and the accumulation order of the atomics differs)
    assert abs(res["ctypes"]["gnorm"] - res["pybind"]["gnorm"]) <= 1e-2 * abs(res["ctypes"]["gnorm"]) and res["ctypes"]["gnorm"] > 0


def test_host_cost_of_a_call_through_either_flavour(both):
    """BASELINE configs[1] (the cow, 256^2, K = 8, coarse + fine forward) per call through the ctypes and through the pybind module:
    the launches are the same five, what differs is the host side (0.130 ms wall for 0.096 ms of kernels).  Printed;
    both must stay launch-bound territory (< 1 ms)."""
    _C, pyb = both
    d = torch.device("cuda:0")
    g = np.load(os.path.join(U.GOLDEN, "cow_ref.npz"))
    fv = torch.from_numpy(g["verts_ndc"])[torch.from_numpy(g["faces"]).long()].contiguous().to(d)
    F = fv.shape[0]
    args = (fv, torch.zeros(1, dtype=torch.int64, device=d), torch.tensor([F], device=d), torch.full((F,), -1, dtype=torch.int64, device=d),
            (256, 256), 1e-4, 8, 16, max(10000, F // 5), True, True, False)

    def wall(mod, iters=300):
        for _ in range(20):
            mod.rasterize_meshes(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            mod.rasterize_meshes(*args)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    t = {"ctypes": wall(_C), "pybind": wall(pyb)}
    t["ctypes_again"], t["pybind_again"] = wall(_C), wall(pyb)
    print("\n[config 2 per call, ms]", {k: round(v, 4) for k, v in t.items()})
    assert 0 < min(t.values()) and max(t.values()) < 1.0
