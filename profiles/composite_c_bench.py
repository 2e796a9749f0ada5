#!/usr/bin/env python
"""Compositor kernel times vs the feature width C on config-4 fragments (1M points, 512x512, K=10).
Run on the GPU box:  python profiles/composite_c_bench.py 3 16 64"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import pytorch3d_amd as p3d
    from pytorch3d_amd import _lib

    d = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    P, H, K, r = 1_000_000, 512, 10, 0.01
    pts = torch.cat([torch.rand(P, 2, generator=gen) * 2 - 1, torch.rand(P, 1, generator=gen) * 2 + 0.5], 1).to(d)
    idx, zbuf, dists = p3d.rasterize_points(p3d.PackedPointclouds([pts]), image_size=H, radius=r, points_per_pixel=K)
    w = (1 - dists.permute(0, 3, 1, 2) / (r * r)).detach().requires_grad_(True)
    il = idx.long().permute(0, 3, 1, 2)
    lib = _lib.load()
    for C in [int(x) for x in sys.argv[1:]] or [3]:
        feats = torch.rand(C, P, generator=gen).to(d).requires_grad_(True)
        g = torch.randn(1, C, H, H, generator=gen).to(d)
        for name, fn in (("alpha", p3d.alpha_composite), ("norm", p3d.norm_weighted_sum)):
            def step():
                feats.grad = w.grad = None
                fn(il, w, feats).backward(g)

            step()
            torch.cuda.synchronize()
            lib.p3d_profile_reset()
            lib.p3d_profile_enable(1)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            lib.p3d_profile_enable(0)
            print(f"C={C} {name}: " + ", ".join(f"{k} {ms / c:.3f} ms" for k, (c, ms) in sorted(_lib.profile_snapshot().items())),
                  flush=True)
        del feats, g


if __name__ == "__main__":
    main()
