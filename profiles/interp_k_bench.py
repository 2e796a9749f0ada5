#!/usr/bin/env python
"""interp_face_attrs backward at K values without a vector-row kernel: flat (64 consecutive samples per step) vs
image-shaped (8x8 pixel tiles, step per k) mapping.  Run on the GPU box:  python profiles/interp_k_bench.py 10 6 3"""
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
    from pytorch3d_amd import _C, _lib

    d = torch.device("cuda:0")
    B = 32
    verts, faces = U.hetero_batch(B, seed=0)
    m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
    blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
    lib = _lib.load()
    gen = torch.Generator().manual_seed(0)
    attrs = torch.rand(m.faces_packed().shape[0], 3, 3, generator=gen).to(d)
    for K in [int(x) for x in sys.argv[1:]] or [10]:
        p2f, _, bary, _ = p3d.rasterize_meshes(m, image_size=512, blur_radius=blur, faces_per_pixel=K,
                                               perspective_correct=True, clip_barycentric_coords=True)
        g = torch.randn(B, 512, 512, K, 3, generator=gen).to(d)
        P = p2f.numel()
        for name, shape in (("flat", None), ("nhwk", (B, 512, 512, K))):
            fn = lambda: _C.interp_face_attrs_backward(p2f.view(-1), bary.view(-1, 3), attrs, g.view(-1, 3), image_shape=shape)
            fn()
            torch.cuda.synchronize()
            lib.p3d_profile_reset()
            lib.p3d_profile_enable(1)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            lib.p3d_profile_enable(0)
            cnt, ms = _lib.profile_snapshot()["interp_bwd"]
            print(f"K={K} {name}: {ms / cnt:.3f} ms  ({P * 44 / (ms / cnt) / 1e6:.0f} GB/s algorithmic)", flush=True)


if __name__ == "__main__":
    main()
