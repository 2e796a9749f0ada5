#!/usr/bin/env python
"""Per-kernel timing of the other BASELINE configs (the headline config 3 is bench.py) on one MI355X.

    python profiles/bench_configs.py            -> one JSON line per config

config 2: one ~5k-face mesh (ico_sphere(4), 5120 faces -- the cow OBJ lives in the reference checkout, which
          does not exist on the GPU box), 256x256, K=8, blur 1e-4, coarse + fine forward
config 4: 1M points, 512x512, K=10, r=0.01, rasterize_points fwd+bwd + alpha compositor fwd+bwd (C=3)
interp  : interpolate_face_attributes fwd+bwd, D=3, on the config-3 fragments (P = 64*512*512*8)
Kernel times are HIP events on the launch stream (p3d_profile_*); GB/s = algorithmic bytes / time.
"""
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
PEAK = 8000.0


def timed(lib, _lib, fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    lib.p3d_profile_reset()
    lib.p3d_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters * 1e3
    lib.p3d_profile_enable(0)
    prof = _lib.profile_snapshot()
    return wall, {k: ms / n for k, (n, ms) in prof.items()}


def main():
    import _util as U
    import pytorch3d_amd as p3d
    from pytorch3d_amd import _C, _lib

    lib = _lib.load()
    d = torch.device("cuda:0")
    out = []

    # ---- config 2 ---------------------------------------------------------------------------------
    v, f = U.ico_sphere(4)
    m = p3d.PackedMeshes([U.to_ndc(v).to(d)], [f.to(d)])
    H = W = 256
    K = 8

    def c2():
        return p3d.rasterize_meshes(m, image_size=H, blur_radius=1e-4, faces_per_pixel=K, perspective_correct=True,
                                    clip_barycentric_coords=True)

    wall, k = timed(lib, _lib, c2, iters=20)
    alg = H * W * K * 28 + f.shape[0] * 44 + 16
    kern = sum(k.values())
    out.append({"config": "2: ico_sphere(4) 5120 faces, 256x256, K=8, blur 1e-4, coarse+fine fwd", "wall_ms": wall,
                "kernels_ms": k, "kernel_sum_ms": kern, "algorithmic_bytes": alg,
                "GBps_over_kernel_sum": alg / kern / 1e6, "Mpix_per_s_wall": H * W / wall / 1e3})

    # ---- config 4 ---------------------------------------------------------------------------------
    gen = torch.Generator().manual_seed(0)
    P = 1_000_000
    pts = torch.cat([torch.rand(P, 2, generator=gen) * 2 - 1, torch.rand(P, 1, generator=gen) * 2 + 0.5], 1).to(d)
    feats = torch.rand(P, 3, generator=gen).to(d)
    H = W = 512
    K = 10
    r = 0.01
    pts_g = pts.clone().requires_grad_(True)
    feats_g = feats.clone().requires_grad_(True)
    g_img = torch.randn(1, 3, H, W, generator=gen).to(d)

    pc = p3d.PackedPointclouds([pts_g])

    def c4():
        pts_g.grad = None
        feats_g.grad = None
        idx, zbuf, dists = p3d.rasterize_points(pc, image_size=H, radius=r, points_per_pixel=K)
        weights = 1 - dists.permute(0, 3, 1, 2) / (r * r)  # points/renderer.py:64-65
        img = p3d.alpha_composite(idx.long().permute(0, 3, 1, 2), weights, feats_g.permute(1, 0))
        img.backward(g_img)
        return idx

    wall, k = timed(lib, _lib, c4, iters=5)
    idx = c4()
    fill = float((idx >= 0).float().mean())
    px = H * W
    alg = {"points_fine": px * K * 12 + P * 16, "points_backward": px * K * 12 + P * 24,
           "alpha_composite_fwd": px * K * 12 + min(px * K, P) * 12 + px * 12,
           "alpha_composite_bwd": px * K * 12 + min(px * K, P) * 12 + px * 12 + px * K * 4 + P * 12}
    out.append({"config": "4: 1M points, 512x512, K=10, r=0.01, rasterize fwd+bwd + alpha composite fwd+bwd (C=3)",
                "wall_ms": wall, "kernels_ms": k, "kernel_sum_ms": sum(k.values()), "slot_fill": fill,
                "algorithmic_bytes": alg,
                "GBps": {n: alg[n] / k[n] / 1e6 for n in alg if n in k},
                "frac_of_hbm_peak": {n: alg[n] / k[n] / 1e6 / PEAK for n in alg if n in k},
                "Mpix_per_s_wall": px / wall / 1e3})

    # ---- interp on config-3 fragments ----------------------------------------------------------------
    verts, faces = U.hetero_batch(64, seed=0)
    m3 = p3d.PackedMeshes([x.to(d) for x in verts], [x.to(d) for x in faces])
    blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
    p2f, zbuf, bary, dists = p3d.rasterize_meshes(m3, image_size=512, blur_radius=blur, faces_per_pixel=8,
                                                  perspective_correct=True, clip_barycentric_coords=True)
    del zbuf, dists
    F = m3.faces_packed().shape[0]
    D = 3
    attrs = torch.rand(F, 3, D, generator=gen).to(d).requires_grad_(True)
    bary_g = bary.detach().requires_grad_(True)
    g_out = torch.randn(64, 512, 512, 8, D, generator=gen).to(d)

    def ci():
        attrs.grad = None
        bary_g.grad = None
        o = p3d.interpolate_face_attributes(p2f, bary_g, attrs)
        o.backward(g_out)

    wall, k = timed(lib, _lib, ci, iters=3, warm=1)
    Pn = p2f.numel()
    # compulsory bytes given the data: samples without a face read pix_to_face only (their barycentrics and upstream
    # gradients are never fetched); every output element is written
    valid_i = int((p2f >= 0).sum())
    alg = {"interp_fwd": Pn * (8 + D * 4) + valid_i * 12 + F * 3 * D * 4,
           "interp_bwd": Pn * (8 + 12) + valid_i * (12 + D * 4) + 2 * F * 3 * D * 4}
    out.append({"config": "interp_face_attrs fwd+bwd, D=3, on config-3 fragments (P=134M)", "wall_ms": wall,
                "kernels_ms": k, "algorithmic_bytes": alg,
                "GBps": {n: alg[n] / k[n] / 1e6 for n in alg if n in k},
                "frac_of_hbm_peak": {n: alg[n] / k[n] / 1e6 / PEAK for n in alg if n in k}})
    # ---- blending on config-3 fragments -------------------------------------------
    del bary_g, g_out, attrs
    from collections import namedtuple

    Frag = namedtuple("Frag", "pix_to_face zbuf dists")
    zbuf3, dists3 = (t.detach() for t in p3d.rasterize_meshes(m3, image_size=512, blur_radius=blur, faces_per_pixel=8,
                                                              perspective_correct=True,
                                                              clip_barycentric_coords=True)[1::2])
    colors = torch.rand(64, 512, 512, 8, 3, generator=gen).to(d).requires_grad_(True)
    dg = dists3.clone().requires_grad_(True)
    zg = zbuf3.clone().requires_grad_(True)
    g_img = torch.randn(64, 512, 512, 4, generator=gen).to(d)
    bp = p3d.BlendParams(1e-4, 1e-4, (1.0, 1.0, 1.0))

    def cb():
        colors.grad = dg.grad = zg.grad = None
        img = p3d.softmax_rgb_blend(colors, Frag(p2f, zg, dg), bp)
        img.backward(g_img)

    wall, k = timed(lib, _lib, cb, iters=3, warm=1)
    px = 64 * 512 * 512
    # compulsory bytes given the data: a pixel without a face is answered from pix_to_face alone (background colour /
    # zero gradient rows), so only covered pixels read their distances, depths and colours (20 B per sample)
    covered = int((p2f >= 0).any(-1).sum())
    alg = {"softmax_blend_fwd": px * (8 * 8 + 16) + covered * 8 * 20,
           "softmax_blend_bwd": px * (8 * 8 + 16 + 8 * 20) + covered * 8 * 20}
    row = {"config": "softmax_rgb_blend fwd+bwd (fused) on config-3 fragments, N=64 512x512 K=8", "wall_ms": wall,
           "covered_pixel_fraction": covered / px,
           "kernels_ms": k, "algorithmic_bytes": alg, "GBps": {n: alg[n] / k[n] / 1e6 for n in alg if n in k},
           "frac_of_hbm_peak": {n: alg[n] / k[n] / 1e6 / PEAK for n in alg if n in k}}

    def torch_blend(nb):
        cs = colors[:nb].detach().requires_grad_(True)
        ds = dg[:nb].detach().requires_grad_(True)
        zs = zg[:nb].detach().requires_grad_(True)
        eps = 1e-10
        mask = p2f[:nb] >= 0
        prob = torch.sigmoid(-ds / 1e-4) * mask
        alpha = torch.prod(1.0 - prob, dim=-1)
        z_inv = (100.0 - zs) / 99.0 * mask
        z_inv_max = torch.max(z_inv, dim=-1).values[..., None].clamp(min=eps)
        wn = prob * torch.exp((z_inv - z_inv_max) / 1e-4)
        delta = torch.exp((eps - z_inv_max) / 1e-4).clamp(min=eps)
        denom = wn.sum(dim=-1)[..., None] + delta
        rgb = ((wn[..., None] * cs).sum(dim=-2) + delta) / denom
        img = torch.cat([rgb, (1.0 - alpha)[..., None]], -1)
        img.backward(g_img[:nb])

    # the reference's formulation (blending.py:147-244 as torch ops + autograd) on the same GPU; at N = 64 torch's
    # own backward fails on ROCm ("invalid configuration argument"), so it is timed on 16 images and scaled by 4
    nb = 16
    torch_blend(nb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        torch_blend(nb)
    torch.cuda.synchronize()
    row["torch_elementwise_same_gpu_ms_scaled_from_16_images"] = (time.perf_counter() - t0) / 3 * 1e3 * (64 / nb)
    out.append(row)
    # ---- Phong shading on config-3 fragments ----------------------------------------
    del colors, dg, zg, g_img
    import pytorch3d_amd.shading as sh

    class _Cam:
        def __init__(self, c):
            self.c = c

        def get_camera_center(self):
            return self.c

    class _Mesh:
        def __init__(self, v, f, n):
            self.v, self.f, self.n = v, f, n

        def verts_packed(self):
            return self.v

        def faces_packed(self):
            return self.f

        def verts_normals_packed(self):
            return self.n

    FragS = namedtuple("FragS", "pix_to_face bary_coords")
    bary3 = p3d.rasterize_meshes(m3, image_size=512, blur_radius=blur, faces_per_pixel=8, perspective_correct=True,
                                 clip_barycentric_coords=True)[2].detach()
    v3 = m3.verts_packed().detach().clone().requires_grad_(True)
    n3 = m3.verts_normals_packed().detach().clone().requires_grad_(True)
    tex3 = torch.rand(64, 512, 512, 8, 3, generator=gen).to(d).requires_grad_(True)
    bary_s = bary3.clone().requires_grad_(True)
    g_col = torch.randn(64, 512, 512, 8, 3, generator=gen).to(d)
    lights = sh.Lights(torch.rand(64, 3, generator=gen).to(d), torch.rand(64, 3, generator=gen).to(d),
                       torch.rand(64, 3, generator=gen).to(d), location=(torch.randn(64, 3, generator=gen) * 2).to(d))
    mats = sh.Materials(torch.rand(1, 3, generator=gen).to(d), torch.rand(1, 3, generator=gen).to(d),
                        torch.rand(1, 3, generator=gen).to(d), torch.tensor([32.0], device=d))
    cam = _Cam((torch.randn(64, 3, generator=gen) - torch.tensor([0.0, 0.0, 3.0])).to(d))
    fp3 = m3.faces_packed()

    def cs():
        v3.grad = n3.grad = tex3.grad = bary_s.grad = None
        col = p3d.phong_shading(_Mesh(v3, fp3, n3), FragS(p2f, bary_s), lights, cam, mats, tex3)
        col.backward(g_col)

    wall, k = timed(lib, _lib, cs, iters=3, warm=1)
    Pn = p2f.numel()
    # per sample: pix_to_face 8 + bary 12 + texel 12 (+ colour 12 out); backward adds grad_colors 12 in and
    # grad_bary 12 + grad_texels 12 out; face records 72 B read (+ 72 B gradient written) per face
    # compulsory bytes given the data: samples without a face read pix_to_face only; their outputs are still written
    valid = int((p2f >= 0).sum())
    alg = {"phong_fwd": Pn * (8 + 12) + valid * (12 + 12) + F * 72,
           "phong_bwd": Pn * (8 + 12 + 12) + valid * (12 + 12 + 12) + 2 * F * 72}
    row = {"config": "phong_shading fwd+bwd (fused) on config-3 fragments, N=64 512x512 K=8, point lights", "wall_ms": wall,
           "kernels_ms": k, "algorithmic_bytes": alg, "GBps": {n: alg[n] / k[n] / 1e6 for n in alg if n in k},
           "frac_of_hbm_peak": {n: alg[n] / k[n] / 1e6 / PEAK for n in alg if n in k}}

    def torch_phong(nb):
        import torch.nn.functional as Fn

        pf, bb = p2f[:nb], bary_s[:nb].detach().requires_grad_(True)
        vv, nn_ = v3.detach().requires_grad_(True), n3.detach().requires_grad_(True)
        tt = tex3[:nb].detach().requires_grad_(True)
        e = lambda t: t[:nb, None, None, None, :]
        pts = p3d.interpolate_face_attributes(pf, bb, vv[fp3])
        nrm = p3d.interpolate_face_attributes(pf, bb, nn_[fp3])
        direction = e(lights.location) - pts
        n_ = Fn.normalize(nrm, p=2, dim=-1, eps=1e-6)
        d_ = Fn.normalize(direction, p=2, dim=-1, eps=1e-6)
        cos = (n_ * d_).sum(-1)
        ldiff = e(lights.diffuse_color) * torch.relu(cos)[..., None]
        mask = (cos > 0).float()
        view = Fn.normalize(e(cam.c) - pts, p=2, dim=-1, eps=1e-6)
        refl = -d_ + 2 * (cos[..., None] * n_)
        alpha = torch.relu((view * refl).sum(-1)) * mask
        lspec = e(lights.specular_color) * torch.pow(alpha, mats.shininess)[..., None]
        col = (e(mats.ambient_color * lights.ambient_color) + mats.diffuse_color * ldiff) * tt + mats.specular_color * lspec
        col.backward(g_col[:nb])

    # the reference's formulation (shading.py:59-96 + lighting.py:17-159 as torch ops + autograd, with THIS package's
    # interpolate_face_attributes kernels inside) on the same GPU, on 8 images scaled by 8
    nb = 8
    torch_phong(nb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        torch_phong(nb)
    torch.cuda.synchronize()
    row["torch_op_chain_same_gpu_ms_scaled_from_8_images"] = (time.perf_counter() - t0) / 3 * 1e3 * (64 / nb)
    out.append(row)
    del v3, n3, tex3, bary_s, g_col, bary3
    # ---- clipping on the config-3 batch ------------------------------------------------
    from pytorch3d_amd import clip as pclip

    fv3 = m3.verts_packed()[m3.faces_packed()].contiguous()
    first3, count3 = m3.mesh_to_faces_packed_first_idx(), m3.num_faces_per_mesh()
    zmid = float(fv3[:, :, 2].median())

    def clip_noop():  # the usual case: everything in front of the plane -> classification + one sync, early exit
        return pclip.clip_faces(fv3, first3, count3, pclip.ClipFrustum(left=-1, right=1, top=-1, bottom=1,
                                                                       perspective_correct=True, z_clip_value=0.1))

    def clip_half():  # stress: the plane cuts through the middle of every mesh
        return pclip.clip_faces(fv3, first3, count3, pclip.ClipFrustum(left=-1, right=1, top=-1, bottom=1,
                                                                       perspective_correct=True, z_clip_value=zmid))

    w0, k0 = timed(lib, _lib, clip_noop, iters=10)
    w1, k1 = timed(lib, _lib, clip_half, iters=10)
    cf = clip_half()
    out.append({"config": f"clip_faces on the config-3 batch ({F} faces): no-op plane / plane through the median depth",
                "noop_wall_ms": w0, "noop_kernels_ms": k0, "half_wall_ms": w1, "half_kernels_ms": k1,
                "half_faces_out": int(cf.face_verts.shape[0]),
                "half_conversion_rows": int(cf.barycentric_conversion.shape[0])})
    for o in out:
        print(json.dumps(o), flush=True)


if __name__ == "__main__":
    main()
