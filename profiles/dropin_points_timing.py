#!/usr/bin/env python
"""BASELINE configs[3] / SURVEY.md 8(d) config 4 AS WRITTEN, through the UNMODIFIED reference classes on the shim:

    PointsRenderer(PointsRasterizer(cameras, PointsRasterizationSettings(512, radius 0.01, points_per_pixel 10, bin_size None)),
                   AlphaCompositor())(Pointclouds(points, features))      loss = sum(image * g)      loss.backward()

(pytorch3d/renderer/points/renderer.py:55-76, points/rasterizer.py:153, points/compositor.py:33; gradients to the points AND
the features).  1M points, xy ~ U(-1,1)^2, z ~ U(0.5,2.5), seed 0, features (P,3) ~ U(0,1); the camera is the identity
orthographic one, so the rasterizer sees exactly these NDC points (PointsRasterizer.transform keeps the view-space z).

    python profiles/dropin_points_timing.py [--steps 20] [--points 1000000]        -> one JSON line (wall + our kernels per step)
    python profiles/dropin_points_timing.py --check                                -> also runs the SAME chain with
        `pytorch3d._C` = the reference's own device kernels (oracle/_ref/p3d_ref_hip_nofma.so) in the same process and compares
        image, fragments and both gradients (tests/test_gpu_points_renderer_dropin.py asserts on the numbers).

bench.py runs it in a subprocess (the shim must not leak into the bench process) as `other_configs.config4_points_renderer_dropin`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HOT = ("rasterize_points", "rasterize_points_backward", "accum_alphacomposite", "accum_alphacomposite_backward")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prewarm-s", type=float, default=0.4,
                    help="seconds of the same step, untimed, ahead of the warm-up steps: bench.py's protocol (a fresh box needs ~0.4 s of "
                         "load before its clocks settle; the few milliseconds of warm-up + timed steps here end before that).  0 = none")
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--image-size", type=int, default=512)
    ap.add_argument("--mode", default="c_only", choices=["c_only", "patched"],
                    help="patched: shim.install(patch_python=True) -- PointsRasterizer.forward transforms the PACKED points in one launch, "
                         "the compositing functions run as one autograd node without clones")
    ap.add_argument("--no-fuse", action="store_true",
                    help="patched mode without the fused PointsRenderer node (pytorch3d_amd.render_points): the operator chain of rounds 4-5")
    ap.add_argument("--check", action="store_true", help="compare with the reference's own device kernels under the same Python")
    ap.add_argument("--graph", action="store_true",
                    help="also: the same step on a Pointclouds built once, eager and captured in a HIP graph (torch.cuda.graph) and replayed -- "
                         "what is left of the step when the host side and the launch gaps are taken out.  EXPERIMENTAL: on this image "
                         "(ROCm 7.2, torch 2.10) capture_end of the forward + backward capture dies with a segmentation fault inside "
                         "the graph instantiation (profiles/r06/c18/err.txt); bench.py does not pass this flag")
    args = ap.parse_args()
    stage = os.path.join(ROOT, "oracle", "_ref", "reference_py")
    ref_root = None
    for cand in (os.environ.get("P3D_REFERENCE_ROOT"), "/root/reference", stage):
        if cand and os.path.isdir(os.path.join(cand, "pytorch3d", "renderer")):
            ref_root = cand
            break
    if ref_root is None:
        print(json.dumps({"value": None, "reason": "the reference's Python package is not on this machine (oracle/_ref/reference_py "
                                                   "is staged by oracle/stage_reference.py in the build container)"}))
        return
    import torch

    import run_reference_suite as rrs  # the iopath / imageio stubs the reference's package needs in this image

    rrs._stub_missing_packages()
    import pytorch3d_amd.shim as shim

    shim.install(ref_root, patch_python=False)
    from pytorch3d.renderer import (AlphaCompositor, FoVOrthographicCameras, PointsRasterizationSettings, PointsRasterizer,
                                    PointsRenderer)
    from pytorch3d.structures import Pointclouds

    if args.mode == "patched":
        shim.patch_reference_python()
        shim.FUSE_POINTS_RENDERER = not args.no_fuse
    from pytorch3d_amd import _lib

    d = torch.device("cuda:0")
    P, H, K, r, C = args.points, args.image_size, 10, 0.01, 3
    gen = torch.Generator().manual_seed(0)
    pts0 = torch.cat([torch.rand(P, 2, generator=gen) * 2 - 1, torch.rand(P, 1, generator=gen) * 2 + 0.5], 1).to(d)
    feats0 = torch.rand(P, C, generator=gen).to(d)
    g_img = torch.randn((1, H, H, C), generator=gen).to(d)
    pts = pts0.clone().requires_grad_(True)
    feats = feats0.clone().requires_grad_(True)
    cams = FoVOrthographicCameras(device=d)
    rs = PointsRasterizationSettings(image_size=H, radius=r, points_per_pixel=K, bin_size=None)
    renderer = PointsRenderer(rasterizer=PointsRasterizer(cameras=cams, raster_settings=rs), compositor=AlphaCompositor())

    def step():
        pts.grad = None
        feats.grad = None
        image = renderer(Pointclouds(points=[pts], features=[feats]))
        (image * g_img).sum().backward()
        return image

    t_pre, n_pre = time.perf_counter(), 0
    while args.prewarm_s > 0 and time.perf_counter() - t_pre < args.prewarm_s:
        image = step()
        torch.cuda.synchronize()
        n_pre += 1
    for _ in range(args.warmup):
        image = step()
    torch.cuda.synchronize()
    lib = _lib.load()
    lib.p3d_profile_reset()
    lib.p3d_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        image = step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.steps * 1e3
    lib.p3d_profile_enable(0)
    kern = {k: round(ms / args.steps, 4) for k, (n, ms) in sorted(_lib.profile_snapshot().items())}
    out = {"mode": args.mode + ("" if args.mode != "patched" or args.no_fuse else " (fused PointsRenderer node)"), "chain": "PointsRenderer(PointsRasterizer, AlphaCompositor) fwd + sum(image*g).backward() to points and features, "
                    "unmodified reference classes over pytorch3d._C = pytorch3d_amd",
           "points": P, "image_size": H, "points_per_pixel": K, "radius": r, "ms_per_step": wall, "steps": args.steps, "prewarm_s": args.prewarm_s, "prewarm_steps": n_pre,
           "Mpix_s": H * H / (wall * 1e-3) / 1e6, "our_kernels_ms_per_step": kern, "our_kernels_sum_ms": round(sum(kern.values()), 4),
           "grad_finite": bool(torch.isfinite(pts.grad).all() and torch.isfinite(feats.grad).all()),
           "covered": float((image.abs().sum(-1) > 0).float().mean()),
           "reference": ref_root if ref_root != stage else "oracle/_ref/reference_py (staged copy)"}
    if args.mode == "patched":
        out["patched_calls"] = {k: v for k, v in shim.PATCH_CALLS.items()}
    if args.graph:
        try:
            want_img, want_gp = image.detach().clone(), pts.grad.clone()
            clouds = Pointclouds(points=[pts], features=[feats])
            clouds.points_packed(), clouds.features_packed(), clouds.cloud_to_packed_first_idx(), clouds.num_points_per_cloud()  # the cloud's caches, built once

            def step_built():
                pts.grad.zero_()
                feats.grad.zero_()
                img = renderer(clouds)
                (img * g_img).sum().backward(retain_graph=True)  # (the cat of the cloud's packed tensors was recorded once, outside)
                return img

            def wall_of(fn, n):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e3

            eager_built = wall_of(step_built, args.steps)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step_built()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                img_g = step_built()
            replay = wall_of(graph.replay, max(args.steps, 50))
            graph.replay()
            torch.cuda.synchronize()
            out["graph"] = {"eager_cloud_built_once_ms": eager_built, "graph_replay_ms": replay,
                            "image_equal_to_eager": bool(torch.equal(img_g.detach(), want_img)),
                            "grad_points_max_rel_diff": float((pts.grad - want_gp).abs().max() / want_gp.abs().max().clamp_min(1e-12)),
                            "note": "forward + backward of the patched chain on a Pointclouds object built once; torch.cuda.graph capture of the whole "
                                    "step (zeroing the .grad tensors included), replayed"}
        except Exception as e:
            out["graph"] = {"error": repr(e)[:400]}
    if args.check and args.mode == "patched":
        # the patched chain against the un-patched one (the reference's own Python over the same `_C`) in this process
        mine = {"image": image.detach().clone(), "gp": pts.grad.clone(), "gf": feats.grad.clone()}
        shim.uninstall_python_patches()
        image_c = step()
        torch.cuda.synchronize()
        out["check"] = {"against": "the same chain with the reference's own Python (patches uninstalled) over the same _C",
                        "image_bit_equal": bool(torch.equal(mine["image"], image_c.detach())),
                        "image_max_abs_diff": float((mine["image"] - image_c.detach()).abs().max()),
                        "grad_points_max_abs_diff": float((mine["gp"] - pts.grad).abs().max()), "grad_points_max_abs": float(pts.grad.abs().max()),
                        "grad_features_max_abs_diff": float((mine["gf"] - feats.grad).abs().max()), "grad_features_max_abs": float(feats.grad.abs().max())}
    if args.check and args.mode == "c_only":
        from oracle import oracle as orc

        mod = orc.ref_hip_module(nofma=True)
        if mod is None:
            out["check"] = {"skipped": "oracle/_ref/p3d_ref_hip_nofma.so not built"}
        else:
            frag = renderer.rasterizer(Pointclouds(points=[pts0], features=[feats0]))
            ours = {"image": image.detach().clone(), "gp": pts.grad.clone(), "gf": feats.grad.clone(), "idx": frag.idx.clone(),
                    "zbuf": frag.zbuf.clone(), "dists": frag.dists.clone()}
            shim_mod = sys.modules["pytorch3d._C"]
            saved = {n: getattr(shim_mod, n) for n in HOT}
            for n in HOT:  # the reference's Python looks the operators up on the module at call time
                setattr(shim_mod, n, getattr(mod, n))
            # The reference's binned device path is not deterministic where depths tie exactly (a bin's points across 512-point chunks
            # arrive in atomicAdd order, rasterize_coarse.cu:185): the fragments compared below are the ones THIS step composited.
            seen = []

            def recording(*a):
                res = mod.rasterize_points(*a)
                seen.append(res)
                return res

            shim_mod.rasterize_points = recording
            try:
                t0 = time.perf_counter()
                image_ref = step()
                torch.cuda.synchronize()
                ref_ms = (time.perf_counter() - t0) * 1e3
            finally:
                for n in HOT:
                    setattr(shim_mod, n, saved[n])
            import collections

            fr = collections.namedtuple("F", "idx zbuf dists")(*seen[-1])
            same = ours["idx"] == fr.idx
            z = ours["zbuf"]
            tie = torch.zeros_like(same)
            tie[..., 1:] |= z[..., 1:] == z[..., :-1]
            tie[..., :-1] |= z[..., :-1] == z[..., 1:]
            tie[..., K - 1] = True
            # Where the two queues kept different points at an exact depth tie (the reference's CUDA queue orders by z alone, ours by
            # (z, idx)) the pixel composites other features: image and gradients are compared on the pixels whose K indices all
            # agree, and on the points that appear in no differing pixel (a pixel's transmittances couple all of its entries).
            bad_pix = (~same).any(-1)  # (1, H, W)
            touched = torch.zeros(P, dtype=torch.bool, device=d)
            for t in (ours["idx"], fr.idx):
                ids = t[bad_pix].reshape(-1).long()
                touched[ids[ids >= 0]] = True
            good = ~bad_pix
            clean = ~touched
            diff_img = (ours["image"] - image_ref.detach()).abs().amax(-1) * good
            wy, wx = divmod(int(diff_img[0].argmax()), H)
            worst = {"pixel": [wy, wx], "ours_idx": ours["idx"][0, wy, wx].tolist(), "ref_idx": fr.idx[0, wy, wx].tolist(),
                     "ours_zbuf": ours["zbuf"][0, wy, wx].tolist(), "ours_dists": ours["dists"][0, wy, wx].tolist(), "ref_dists": fr.dists[0, wy, wx].tolist(),
                     "ours_image": ours["image"][0, wy, wx].tolist(), "ref_image": image_ref[0, wy, wx].tolist()}
            out["check"] = {
                "against": "oracle/_ref/p3d_ref_hip_nofma.so (the reference's .cu files for gfx950, -ffp-contract=off) under the same Python",
                "reference_step_ms_single_run": ref_ms,
                "zbuf_bit_equal": bool(torch.equal(ours["zbuf"].view(torch.int32), fr.zbuf.view(torch.int32))),
                "idx_differences": int((~same).sum()), "idx_differences_not_at_exact_depth_ties": int((~same & ~tie).sum()),
                "idx_entries": same.numel(), "pixels_with_an_idx_difference": int(bad_pix.sum()), "points_in_those_pixels": int(touched.sum()),
                "dists_bit_equal_where_idx_agrees": bool(torch.equal(ours["dists"].view(torch.int32)[same], fr.dists.view(torch.int32)[same])),
                "image_max_abs_diff": float((ours["image"] - image_ref.detach())[good].abs().max()),
                "image_max_abs_diff_all_pixels": float((ours["image"] - image_ref.detach()).abs().max()),
                "grad_points_max_abs_diff": float((ours["gp"] - pts.grad)[clean].abs().max()), "grad_points_max_abs": float(pts.grad.abs().max()),
                "grad_features_max_abs_diff": float((ours["gf"] - feats.grad)[clean].abs().max()), "grad_features_max_abs": float(feats.grad.abs().max()),
                "points_with_gradient": [int((ours["gp"] != 0).any(1).sum()), int((pts.grad != 0).any(1).sum())],
                "worst_good_pixel": worst}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
