#!/usr/bin/env python
"""rasterize_points forward / backward kernel times vs points_per_pixel on config 4 (1M points, 512x512, r=0.01).
Run on the GPU box:  python profiles/points_k_sweep.py 1 8 10 16 32 50"""
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
    P, H, r = 1_000_000, 512, 0.01
    pts = torch.cat([torch.rand(P, 2, generator=gen) * 2 - 1, torch.rand(P, 1, generator=gen) * 2 + 0.5], 1).to(d)
    pts.requires_grad_(True)
    pc = p3d.PackedPointclouds([pts])
    lib = _lib.load()
    for K in [int(x) for x in sys.argv[1:]] or [10]:
        gz = torch.randn(1, H, H, K, generator=gen).to(d)
        gd = torch.randn(1, H, H, K, generator=gen).to(d)

        def step():
            pts.grad = None
            idx, z, dist = p3d.rasterize_points(pc, image_size=H, radius=r, points_per_pixel=K)
            torch.autograd.backward([z, dist], [gz, gd])
            return idx

        idx = step()
        torch.cuda.synchronize()
        lib.p3d_profile_reset()
        lib.p3d_profile_enable(1)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        lib.p3d_profile_enable(0)
        pr = _lib.profile_snapshot()
        print(f"K={K}: points_fine {pr['points_fine'][1] / pr['points_fine'][0]:.3f} ms, points_backward "
              f"{pr['points_backward'][1] / pr['points_backward'][0]:.3f} ms, slot fill {float((idx >= 0).float().mean()):.3f}",
              flush=True)


if __name__ == "__main__":
    main()
