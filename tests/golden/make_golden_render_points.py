"""End-to-end golden for the point chain rasterize_points -> alpha / norm-weighted compositing, generated FROM THE
REFERENCE (build container only):

    python tests/golden/make_golden_render_points.py   ->  tests/golden/render_points_ref.npz

The reference's own `PointsRenderer(PointsRasterizer, AlphaCompositor | NormWeightedCompositor)`
(renderer/points/renderer.py:30-72, rasterizer.py:95-170, compositor.py) with FoVPerspectiveCameras on CPU, plus torch
autograd of a random loss to the point features and the point positions in NDC (the camera transform, out of scope here,
is bypassed by handing the rasterizer points that already are in NDC with an identity FoVOrthographic camera? -- no:
the fixture stores the NDC points the rasterizer derives and the gradient wrt the world points is not recorded).
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    import make_golden as mg

    mg.bind_reference()
    from pytorch3d.renderer import (AlphaCompositor, FoVPerspectiveCameras, NormWeightedCompositor, PointsRasterizationSettings,
                                    PointsRasterizer, PointsRenderer, look_at_view_transform)
    from pytorch3d.structures import Pointclouds

    gen = torch.Generator().manual_seed(21)
    pts_l = [torch.randn(700, 3, generator=gen) * 0.5, torch.randn(450, 3, generator=gen) * 0.4]
    feat_l = [torch.rand(p.shape[0], 4, generator=gen).requires_grad_(True) for p in pts_l]
    clouds = Pointclouds(points=pts_l, features=feat_l)
    R, T = look_at_view_transform(dist=2.5, elev=torch.tensor([5.0, 40.0]), azim=torch.tensor([15.0, -70.0]))
    cameras = FoVPerspectiveCameras(R=R, T=T, znear=0.1)
    settings = PointsRasterizationSettings(image_size=(40, 56), radius=0.04, points_per_pixel=7, bin_size=0)
    rasterizer = PointsRasterizer(cameras=cameras, raster_settings=settings)
    out = {}
    frag = rasterizer(clouds)
    out.update(points_ndc=rasterizer.transform(clouds).points_packed(), idx=frag.idx, zbuf=frag.zbuf, dists=frag.dists,
               features=torch.cat(feat_l), num_points=torch.tensor([p.shape[0] for p in pts_l]), radius=0.04, K=7,
               image_size=torch.tensor([40, 56]))
    for tag, comp in (("alpha", AlphaCompositor(background_color=(0.1, 0.2, 0.3, 1.0))), ("norm", NormWeightedCompositor())):
        for f in feat_l:
            f.grad = None
        img = PointsRenderer(rasterizer=rasterizer, compositor=comp)(clouds)
        g = torch.randn(img.shape, generator=gen)
        (img * g).sum().backward()
        out.update({f"{tag}_image": img, f"{tag}_grad_image": g, f"{tag}_grad_features": torch.cat([f.grad for f in feat_l])})
    mg.save("render_points_ref", **out)
    print("coverage", float((frag.idx[..., 0] >= 0).float().mean()))


if __name__ == "__main__":
    main()
