#!/usr/bin/env python
"""Kernel times of the fused Phong shading on the config-3 fragments (run on the GPU box).

    python profiles/shade_bench.py [vcol]      texels given per sample (D = 6) or vertex colours fused in (D = 9)
"""
import math
import os
import sys
from collections import namedtuple

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class Cam:
    def __init__(self, c):
        self.c = c

    def get_camera_center(self):
        return self.c


class Mesh:
    def __init__(self, v, f, n):
        self.v, self.f, self.n = v, f, n

    def verts_packed(self):
        return self.v

    def faces_packed(self):
        return self.f

    def verts_normals_packed(self):
        return self.n


def main():
    import _util as U
    import pytorch3d_amd as p3d
    import pytorch3d_amd.shading as sh
    from pytorch3d_amd import _lib

    vcol = len(sys.argv) > 1 and sys.argv[1] == "vcol"
    B = int(os.environ.get("ABL_BATCH", "64"))
    d = torch.device("cuda:0")
    verts, faces = U.hetero_batch(B, seed=0)
    m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
    blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
    p2f, _, bary, _ = p3d.rasterize_meshes(m, image_size=512, blur_radius=blur, faces_per_pixel=8,
                                           perspective_correct=True, clip_barycentric_coords=True)
    gen = torch.Generator().manual_seed(1)
    Frag = namedtuple("Frag", "pix_to_face bary_coords")
    v = m.verts_packed().detach().clone().requires_grad_(True)
    n = m.verts_normals_packed().detach().clone().requires_grad_(True)
    b = bary.detach().clone().requires_grad_(True)
    tex = (torch.rand(v.shape[0], 3, generator=gen) if vcol else torch.rand(B, 512, 512, 8, 3, generator=gen)).to(d)
    tex.requires_grad_(True)
    g = torch.randn(B, 512, 512, 8, 3, generator=gen).to(d)
    L = sh.Lights(torch.rand(B, 3, generator=gen).to(d), torch.rand(B, 3, generator=gen).to(d),
                  torch.rand(B, 3, generator=gen).to(d), location=(torch.randn(B, 3, generator=gen) * 2).to(d))
    M = sh.Materials(torch.rand(1, 3, generator=gen).to(d), torch.rand(1, 3, generator=gen).to(d),
                     torch.rand(1, 3, generator=gen).to(d), torch.tensor([32.0], device=d))
    cam = Cam((torch.randn(B, 3, generator=gen) - torch.tensor([0.0, 0.0, 3.0])).to(d))
    fn = p3d.phong_shading_vertex_colors if vcol else p3d.phong_shading
    lib = _lib.load()

    def step():
        v.grad = n.grad = b.grad = tex.grad = None
        fn(Mesh(v, m.faces_packed(), n), Frag(p2f, b), L, cam, M, tex).backward(g)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    lib.p3d_profile_reset()
    lib.p3d_profile_enable(1)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    lib.p3d_profile_enable(0)
    print(", ".join(f"{k} {ms / cnt:.3f} ms" for k, (cnt, ms) in sorted(_lib.profile_snapshot().items()) if k.startswith("phong")),
          flush=True)

if __name__ == "__main__":
    main()
