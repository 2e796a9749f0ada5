#!/usr/bin/env python
"""interpolate_face_attributes forward + backward kernel times vs the attribute width D on 16 images of config-3
fragments (512x512, K=8).  Run on the GPU box:  python profiles/interp_d_bench.py 3 4 8 16 32"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import _util as U
    import pytorch3d_amd as p3d
    from pytorch3d_amd import _lib

    d = torch.device("cuda:0")
    B = 16
    verts, faces = U.hetero_batch(B, seed=0)
    m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
    blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
    p2f, _, bary, _ = p3d.rasterize_meshes(m, image_size=512, blur_radius=blur, faces_per_pixel=8,
                                           perspective_correct=True, clip_barycentric_coords=True)
    lib = _lib.load()
    gen = torch.Generator().manual_seed(0)
    F = m.faces_packed().shape[0]
    P = p2f.numel()
    for D in [int(x) for x in sys.argv[1:]] or [3]:
        attrs = torch.rand(F, 3, D, generator=gen).to(d).requires_grad_(True)
        b = bary.detach().clone().requires_grad_(True)
        g = torch.randn(B, 512, 512, 8, D, generator=gen).to(d)

        def step():
            attrs.grad = b.grad = None
            p3d.interpolate_face_attributes(p2f, b, attrs).backward(g)

        step()
        torch.cuda.synchronize()
        lib.p3d_profile_reset()
        lib.p3d_profile_enable(1)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        lib.p3d_profile_enable(0)
        pr = _lib.profile_snapshot()
        fw, bw = pr["interp_fwd"][1] / pr["interp_fwd"][0], pr["interp_bwd"][1] / pr["interp_bwd"][0]
        af, ab = P * (20 + 4 * D), P * (32 + 4 * D)
        print(f"D={D}: interp_fwd {fw:.3f} ms ({af / fw / 1e6:.0f} GB/s), interp_bwd {bw:.3f} ms ({ab / bw / 1e6:.0f} GB/s)", flush=True)
        del attrs, b, g


if __name__ == "__main__":
    main()
