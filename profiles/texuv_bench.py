#!/usr/bin/env python
"""Fused UV texture sampling vs the reference's formulation (interpolate + lerp + F.grid_sample on K expanded NCHW map
copies, textures.py:1190-1268) on config-3 fragments: B images 512x512 K=8, one 1024x1024x3 map per mesh, fwd+bwd.
Run on the GPU box:  python profiles/texuv_bench.py"""
import math
import os
import sys
import time
from collections import namedtuple

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import _util as U
    import pytorch3d_amd as p3d
    from pytorch3d_amd import _lib

    B = int(os.environ.get("ABL_BATCH", "16"))
    d = torch.device("cuda:0")
    verts, faces = U.hetero_batch(B, seed=0)
    m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
    blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
    p2f, _, bary, _ = p3d.rasterize_meshes(m, image_size=512, blur_radius=blur, faces_per_pixel=8,
                                           perspective_correct=True, clip_barycentric_coords=True)
    gen = torch.Generator().manual_seed(0)
    F = m.faces_packed().shape[0]
    if os.environ.get("TEXUV_RANDOM"):  # a random uv per face corner: no texel locality at all (worst case for the map atomics)
        fu = torch.rand(F, 3, 2, generator=gen).to(d).requires_grad_(True)
    else:  # a planar chart: uv = the vertex's NDC xy mapped to [0, 1] -- neighbouring pixels read neighbouring texels
        uvv = (m.verts_packed()[:, :2] * 0.45 + 0.5).clamp(0, 1)
        fu = uvv[m.faces_packed()].contiguous().requires_grad_(True)
    maps = torch.rand(B, 1024, 1024, 3, generator=gen).to(d).requires_grad_(True)
    b = bary.detach().clone().requires_grad_(True)
    g = torch.randn(B, 512, 512, 8, 3, generator=gen).to(d)
    Frag = namedtuple("Frag", "pix_to_face bary_coords")
    lib = _lib.load()

    def fused():
        fu.grad = maps.grad = b.grad = None
        p3d.sample_textures_uv(Frag(p2f, b), fu, maps).backward(g)

    def torch_chain():
        fu.grad = maps.grad = b.grad = None
        N, H, W, K = p2f.shape
        uv = p3d.interpolate_face_attributes(p2f, b, fu)
        uv = uv.permute(0, 3, 1, 2, 4).reshape(N * K, H, W, 2)
        tm = maps.permute(0, 3, 1, 2)[None].expand(K, -1, -1, -1, -1).transpose(0, 1).reshape(N * K, 3, 1024, 1024)
        uv = torch.lerp(uv.new_tensor([-1.0, 1.0]), uv.new_tensor([1.0, -1.0]), uv)
        t = torch.nn.functional.grid_sample(tm, uv, mode="bilinear", align_corners=True, padding_mode="border")
        t.reshape(N, K, 3, H, W).permute(0, 3, 4, 1, 2).backward(g)

    def wall(fn, it=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(it):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / it * 1e3

    fused()
    torch.cuda.synchronize()
    lib.p3d_profile_reset()
    lib.p3d_profile_enable(1)
    wf = wall(fused)
    lib.p3d_profile_enable(0)
    k = {n: ms / c for n, (c, ms) in _lib.profile_snapshot().items() if n.startswith("sample_uv")}
    wt = wall(torch_chain, it=3)
    P = p2f.numel()
    print(f"B={B}: fused wall {wf:.3f} ms (kernels {k}), torch chain wall {wt:.3f} ms; samples {P / 1e6:.1f} M", flush=True)


if __name__ == "__main__":
    main()
