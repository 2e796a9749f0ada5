#!/usr/bin/env python
"""Cases of the patched PointsRenderer.forward (pytorch3d_amd/shim.py: the fused node of pytorch3d_amd.render_points and its fall-backs),
run in a process of its own (the shim replaces sys.modules entries): every case renders with the patches installed, fused node on and
off, and with the reference's own Python over the same `_C`; prints one JSON line {case: {...}}.  tests/test_gpu_points_renderer_dropin.py
asserts on it."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    stage = os.path.join(ROOT, "oracle", "_ref", "reference_py")
    ref_root = next((c for c in (os.environ.get("P3D_REFERENCE_ROOT"), "/root/reference", stage)
                     if c and os.path.isdir(os.path.join(c, "pytorch3d", "renderer"))), None)
    if ref_root is None:
        print(json.dumps({"skipped": "the reference's Python package is not on this machine"}))
        return
    import torch

    import run_reference_suite as rrs

    rrs._stub_missing_packages()
    import pytorch3d_amd.shim as shim

    shim.install(ref_root, patch_python=False)
    from pytorch3d.renderer import (AlphaCompositor, FoVOrthographicCameras, FoVPerspectiveCameras, NormWeightedCompositor,
                                    PointsRasterizationSettings, PointsRasterizer, PointsRenderer)
    from pytorch3d.structures import Pointclouds

    d = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(3)

    def cloud(P, C):
        pts = torch.cat([torch.rand(P, 2, generator=gen) * 2 - 1, torch.rand(P, 1, generator=gen) * 2 + 0.5], 1)
        return pts.to(d), torch.rand(P, C, generator=gen).to(d)

    def case(sizes, C, K, radius, compositor, cams, image_size=64, padded=False, **kw):
        data = [cloud(P, C) for P in sizes]
        g = torch.randn((len(sizes), image_size, image_size, C), generator=gen).to(d)
        rs = PointsRasterizationSettings(image_size=image_size, radius=radius, points_per_pixel=K)
        renderer = PointsRenderer(rasterizer=PointsRasterizer(cameras=cams, raster_settings=rs), compositor=compositor)

        def run():
            pts = [p.clone().requires_grad_(True) for p, _ in data]
            fts = [f.clone().requires_grad_(True) for _, f in data]
            if padded:  # a cloud built from padded tensors: no lists to read the packed tensors from
                pc = Pointclouds(points=torch.stack(pts), features=torch.stack(fts))
            else:
                pc = Pointclouds(points=pts, features=fts)
            img = renderer(pc, **kw)
            (img * g).sum().backward()
            return img.detach(), torch.cat([p.grad for p in pts]), torch.cat([f.grad for f in fts])

        shim.patch_reference_python()
        shim.PATCH_CALLS.clear()
        shim.FUSE_POINTS_RENDERER = True
        fused = run()
        calls = {k: list(v) for k, v in shim.PATCH_CALLS.items() if k.startswith("PointsRenderer")}
        shim.FUSE_POINTS_RENDERER = False
        chain = run()
        shim.uninstall_python_patches()
        ref = run()
        shim.FUSE_POINTS_RENDERER = True

        def dev(a, b):
            return float((a - b).abs().max()), float(b.abs().max())

        return {"calls": calls, "image_equal_to_operator_chain": bool(torch.equal(fused[0], chain[0])),
                "image_vs_reference_python": dev(fused[0], ref[0]), "grad_points_vs_reference_python": dev(fused[1], ref[1]),
                "grad_features_vs_reference_python": dev(fused[2], ref[2]), "grad_points_vs_chain": dev(fused[1], chain[1]),
                "grad_features_vs_chain": dev(fused[2], chain[2]), "covered": float((fused[0].abs().sum(-1) > 0).float().mean())}

    ortho = FoVOrthographicCameras(device=d)
    persp = FoVPerspectiveCameras(device=d)
    out = {
        "one_cloud_rgb": case([3000], 3, 10, 0.05, AlphaCompositor(), ortho),
        "three_ragged_clouds_rgba_perspective": case([2000, 500, 3500], 4, 8, 0.06, AlphaCompositor(), persp),
        "background_color": case([1500, 1500], 3, 6, 0.05, AlphaCompositor(background_color=(0.2, 0.5, 0.9)), ortho),
        "background_color_kwarg_rgba": case([1500], 4, 6, 0.05, AlphaCompositor(), ortho, background_color=(0.1, 0.2, 0.3, 1.0)),
        "padded_cloud": case([1000, 1000], 3, 8, 0.05, AlphaCompositor(), ortho, padded=True),
        "one_channel": case([2500], 1, 16, 0.08, AlphaCompositor(), ortho),
        # not taken by the fused node: the operator chain's own patches (or the reference) run
        "fallback_k_above_16": case([2500], 3, 20, 0.08, AlphaCompositor(), ortho),
        "fallback_five_channels": case([2500], 5, 8, 0.05, AlphaCompositor(), ortho),
        "norm_weighted_compositor": case([2500, 800], 3, 8, 0.05, NormWeightedCompositor(), ortho),
        "norm_weighted_background": case([2500], 4, 10, 0.05, NormWeightedCompositor(background_color=(0.3, 0.3, 0.3, 1.0)), persp),
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
