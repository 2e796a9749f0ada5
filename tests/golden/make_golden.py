"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE (run in the build container only).

    python tests/golden/make_golden.py

Imports the unmodified reference Python package from /root/reference with `pytorch3d._C` bound to
the reference's own CPU kernels (oracle/_ref/p3d_ref_cpu.so, built from the reference sources by
oracle/build.py).  Nothing of this repository's implementation is involved in producing the
vectors:
  * mesh_py_*.npz    rasterize_meshes_python (pytorch3d/renderer/mesh/rasterize_meshes.py:404-619)
                     forward, and torch-autograd gradients of the reference's gradient-check loss
  * mesh_cpp_*.npz   RasterizeMeshesNaiveCpu / RasterizeMeshesBackwardCpu (rasterize_meshes_cpu.cpp)
  * points_*.npz     rasterize_points_python and RasterizePointsNaiveCpu / BackwardCpu
  * composite_*.npz  alphaComposite / weightedSumNorm / weightedSum Cpu forward + backward
  * interp_*.npz     interpolate_face_attributes_python + autograd
The fixtures are small (a few hundred kB in total) and travel to the GPU box, where
/root/reference does not exist.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REFERENCE = "/root/reference"


def bind_reference():
    from oracle import oracle as orc

    ref = orc.ref_module()
    assert ref is not None, "build oracle/_ref first (python oracle/build.py)"
    sys.modules["pytorch3d._C"] = ref
    sys.path.insert(0, REFERENCE)
    import pytorch3d  # noqa: F401

    pytorch3d._C = ref
    return ref


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: tuple(v.shape) for k, v in out.items() if hasattr(v, "shape") and v.ndim})


def main():
    ref = bind_reference()
    import _util as U
    from pytorch3d.ops.interp_face_attrs import interpolate_face_attributes_python
    from pytorch3d.renderer.mesh.rasterize_meshes import rasterize_meshes_python
    from pytorch3d.renderer.points.rasterize_points import rasterize_points_python
    from pytorch3d.structures import Meshes, Pointclouds

    torch.manual_seed(231)
    gen = torch.Generator().manual_seed(231)

    # ---- meshes, reference Python implementation (+ autograd) -------------------------------
    cases = [
        ("a", (20, 20), 0.0, 2, False, False, False),
        ("b", (18, 18), 0.01, 3, True, False, False),
        ("c", (16, 28), 0.01, 3, True, True, False),
        ("d", (22, 14), 0.002, 4, False, True, True),
    ]
    for tag, size, blur, K, persp, clip, cull in cases:
        v1, f1 = U.ico_sphere(0)
        v2, f2 = U.torus(0.3, 0.7, 6, 8)
        verts = [U.to_ndc(v1 * 0.9).clone().requires_grad_(True), U.to_ndc(v2).clone().requires_grad_(True)]
        faces = [f1, f2]
        meshes = Meshes(verts=verts, faces=faces)
        out = rasterize_meshes_python(meshes, size, blur, K, persp, clip, cull, cull_to_frustum=False)
        p2f, zbuf, bary, dists = out
        gz = torch.randn(zbuf.shape, generator=gen)
        gb = torch.randn(bary.shape, generator=gen)
        gd = torch.randn(dists.shape, generator=gen)
        loss = (zbuf * gz).sum() + (bary * gb).sum() + (dists * gd).sum()
        loss.backward()
        save("mesh_py_" + tag, verts0=verts[0], verts1=verts[1], faces0=f1, faces1=f2, image_size=size, blur=blur, K=K,
             persp=persp, clip=clip, cull=cull, pix_to_face=p2f, zbuf=zbuf, bary=bary, dists=dists, grad_zbuf=gz,
             grad_bary=gb, grad_dists=gd, grad_verts0=verts[0].grad, grad_verts1=verts[1].grad)

    # ---- meshes, reference C++ CPU kernels ----------------------------------------------------
    for tag, size, blur, K, persp, clip, cull in [("a", (64, 64), 1e-3, 8, True, True, False),
                                                  ("b", (48, 80), 0.0, 4, False, False, True),
                                                  ("c", (33, 33), 0.01, 2, True, False, False)]:
        verts, faces = U.hetero_batch(3, seed=17, fmin=150, fmax=500)
        fv = torch.cat([v[f] for v, f in zip(verts, faces)], 0)
        cnt = torch.tensor([f.shape[0] for f in faces])
        first = torch.cumsum(cnt, 0) - cnt
        nbr = torch.full((fv.shape[0],), -1, dtype=torch.int64)
        p2f, zbuf, bary, dists = ref._rasterize_meshes_naive(fv, first, cnt, nbr, size, blur, K, persp, clip, cull)
        gz = torch.randn(zbuf.shape, generator=gen)
        gb = torch.randn(bary.shape, generator=gen)
        gd = torch.randn(dists.shape, generator=gen)
        gfv = ref.rasterize_meshes_backward(fv, p2f, gz, gb, gd, persp, clip)
        save("mesh_cpp_" + tag, face_verts=fv, first=first, count=cnt, image_size=size, blur=blur, K=K, persp=persp,
             clip=clip, cull=cull, pix_to_face=p2f, zbuf=zbuf, bary=bary, dists=dists, grad_zbuf=gz, grad_bary=gb,
             grad_dists=gd, grad_face_verts=gfv)

    # ---- points ---------------------------------------------------------------------------------
    for tag, size, K in [("a", (16, 16), 3), ("b", (12, 20), 5)]:
        pts = [torch.cat([torch.rand(60, 2, generator=gen) * 2.2 - 1.1, torch.rand(60, 1, generator=gen) * 2 - 0.2], 1),
               torch.cat([torch.rand(90, 2, generator=gen) * 2.2 - 1.1, torch.rand(90, 1, generator=gen) * 2 - 0.2], 1)]
        pts = [p.clone().requires_grad_(True) for p in pts]
        clouds = Pointclouds(points=pts)
        idx, zbuf, dists = rasterize_points_python(clouds, size, 0.15, K)
        gz = torch.randn(zbuf.shape, generator=gen)
        gd = torch.randn(dists.shape, generator=gen)
        ((zbuf * gz).sum() + (dists * gd).sum()).backward()
        save("points_py_" + tag, points0=pts[0], points1=pts[1], image_size=size, radius=0.15, K=K, idx=idx, zbuf=zbuf,
             dists=dists, grad_zbuf=gz, grad_dists=gd, grad_points0=pts[0].grad, grad_points1=pts[1].grad)
    P = 3000
    pts = torch.cat([torch.rand(P, 2, generator=gen) * 2.4 - 1.2, torch.rand(P, 1, generator=gen) * 2.2 - 0.2], 1)
    first = torch.tensor([0, 1000])
    cnt = torch.tensor([1000, 2000])
    radius = torch.rand(P, generator=gen) * 0.06 + 0.01
    idx, zbuf, dists = ref._rasterize_points_naive(pts, first, cnt, (64, 48), radius, 10)
    gz = torch.randn(zbuf.shape, generator=gen)
    gd = torch.randn(dists.shape, generator=gen)
    gp = ref.rasterize_points_backward(pts, idx, gz, gd)
    save("points_cpp_a", points=pts, first=first, count=cnt, radius=radius, image_size=(64, 48), K=10, idx=idx,
         zbuf=zbuf, dists=dists, grad_zbuf=gz, grad_dists=gd, grad_points=gp)

    # ---- compositors ------------------------------------------------------------------------------
    N, C, Pn, K, H, W = 2, 4, 50, 5, 9, 7
    feat = torch.rand(C, Pn, generator=gen)
    alphas = torch.rand(N, K, H, W, generator=gen)
    pidx = torch.randint(-1, Pn, (N, K, H, W), generator=gen)
    go = torch.randn(N, C, H, W, generator=gen)
    arrays = dict(features=feat, alphas=alphas, points_idx=pidx, grad_out=go)
    for name in ("alphacomposite", "weightedsumnorm", "weightedsum"):
        arrays[name] = getattr(ref, "accum_" + name)(feat, alphas, pidx)
        gf, ga = getattr(ref, "accum_" + name + "_backward")(go, feat, alphas, pidx)
        arrays[name + "_grad_features"] = gf
        arrays[name + "_grad_alphas"] = ga
    save("composite_cpp", **arrays)

    # ---- interpolate_face_attributes ---------------------------------------------------------------
    N, H, W, K, F, D = 2, 5, 6, 3, 12, 4
    p2f = torch.randint(-1, F, (N, H, W, K), generator=gen)
    bary = torch.rand(N, H, W, K, 3, generator=gen).requires_grad_(True)
    attrs = torch.randn(F, 3, D, generator=gen).requires_grad_(True)
    out = interpolate_face_attributes_python(p2f, bary, attrs)
    g = torch.randn(out.shape, generator=gen)
    (out * g).sum().backward()
    save("interp_py", pix_to_face=p2f, bary=bary, face_attrs=attrs, out=out, grad_out=g, grad_bary=bary.grad,
         grad_face_attrs=attrs.grad)


if __name__ == "__main__":
    main()
