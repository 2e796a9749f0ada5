#!/usr/bin/env python
"""End-to-end soft Phong rendering step on the bench workload (BASELINE configs[2] batch: 64 meshes, 512x512, K=8):

    rasterize_meshes -> phong_shading (vertex colours) -> softmax_rgb_blend -> loss -> backward to the vertices

once with the shading and the blend as two operators (round 2) and once with csrc/soft_phong.hip (round 3: one kernel
each way, the per-sample colours never in HBM),

i.e. what MeshRenderer(MeshRasterizer, SoftPhongShader) runs per training step (renderer/mesh/renderer.py:41-63,
shader.py: SoftPhongShader.forward), every stage through this package's fused kernels.  Prints one JSON line with the
wall time per step and the per-kernel HIP-event times.  Run on the GPU box:  python profiles/bench_pipeline.py
"""
import json
import math
import os
import sys
import time
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


def main():
    import _util as U
    import pytorch3d_amd as p3d
    import pytorch3d_amd.shading as sh
    from pytorch3d_amd import _lib

    B, H, K = int(os.environ.get("ABL_BATCH", "64")), 512, 8
    d = torch.device("cuda:0")
    verts, faces = U.hetero_batch(B, seed=0)
    meshes = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
    blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
    gen = torch.Generator().manual_seed(2)
    vp = meshes.verts_packed().clone().requires_grad_(True)
    vcol = torch.rand(vp.shape[0], 3, generator=gen).to(d).requires_grad_(True)
    L = sh.Lights(torch.full((1, 3), 0.5, device=d), torch.full((1, 3), 0.3, device=d), torch.full((1, 3), 0.2, device=d),
                  location=torch.tensor([[0.0, 1.0, -2.0]], device=d))
    M = sh.Materials(torch.ones(1, 3, device=d), torch.ones(1, 3, device=d), torch.ones(1, 3, device=d),
                     torch.tensor([64.0], device=d))
    cam = Cam(torch.tensor([[0.0, 0.0, -2.7]], device=d))
    g_img = torch.randn(B, H, H, 4, generator=gen).to(d)
    Frag = namedtuple("Frag", "pix_to_face zbuf bary_coords dists")
    bp = p3d.BlendParams(1e-4, 1e-4, (1.0, 1.0, 1.0))
    lib = _lib.load()

    def step(fused):
        vp.grad = vcol.grad = None
        m = meshes.update_verts_packed(vp)
        frag = Frag(*p3d.rasterize_meshes(m, image_size=H, blur_radius=blur, faces_per_pixel=K, perspective_correct=True,
                                          clip_barycentric_coords=True))
        if fused:
            img = sh.soft_phong_shading(m, frag, L, cam, M, None, bp, verts_colors_packed=vcol)
        else:
            colors = p3d.phong_shading_vertex_colors(m, frag, L, cam, M, vcol)
            img = p3d.softmax_rgb_blend(colors, frag, bp)
        img.backward(g_img)

    out = {}
    for fused in (False, True):
        for _ in range(3):
            step(fused)
        torch.cuda.synchronize()
        lib.p3d_profile_reset()
        lib.p3d_profile_enable(1)
        iters = 10
        t0 = time.perf_counter()
        for _ in range(iters):
            step(fused)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / iters * 1e3
        lib.p3d_profile_enable(0)
        k = {n: ms / c for n, (c, ms) in sorted(_lib.profile_snapshot().items())}
        out["fused_soft_phong" if fused else "phong_then_blend"] = {
            "wall_ms": wall, "Mpix_per_s": B * H * H / wall / 1e3, "kernel_sum_ms": sum(k.values()),
            "kernels_ms": {n: round(v, 4) for n, v in k.items()}}
    out["config"] = f"soft Phong render step fwd+bwd: rasterize -> Phong (vertex colours) -> softmax blend, N={B} 512x512 K=8"
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
