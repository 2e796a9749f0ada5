#!/usr/bin/env python
"""What a user of the UNMODIFIED reference classes gets through the shim, timed on the bench batch (BASELINE configs[2]).

    python profiles/dropin_timing.py [--mode c_only|patched] [--steps 10]   ->  one JSON line

The reference's own `pytorch3d.structures.Meshes`, `FoVOrthographicCameras`, `RasterizationSettings` and
`MeshRasterizer.forward` (pytorch3d/renderer/mesh/rasterizer.py:219-276; the staged copy under oracle/_ref/reference_py or
P3D_REFERENCE_ROOT) with `pytorch3d._C` provided by pytorch3d_amd.shim, driven the way the reference's tutorials drive it:
`mesh0.offset_verts(deform)` every step, forward, backward to `deform` with the bench's upstream gradients for zbuf / bary
/ dists.  The camera is the identity orthographic one, so the rasterizer sees exactly the bench's NDC geometry;
perspective_correct=True as in the bench, which also switches the reference's near-plane clipping on (z_clip =
znear / 2, rasterizer.py:244-251): nothing is cut, but `clip_faces` runs.

  c_only   `_C` alone is ours: the face gather `verts_packed[faces_packed]` + its index_put_ backward, `clip_faces` with its
           host syncs and the padded camera transform run as the reference's torch code;
  patched  shim.install(patch_python=True): `MeshRasterizer.forward` transforms the PACKED vertices in one launch and calls
           the fused `rasterize_meshes` (gather, clipping, rasterizer, backward to the vertices); `Meshes.offset_verts` and
           the cameras' matrix bookkeeping stay the reference's (measured: ~4.7 ms + ~1 ms of the step, CPU-side).

bench.py runs both in subprocesses (the shim must not leak into the bench process) and reports them next to the mirror's
step time as `dropin`.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="c_only", choices=["c_only", "patched"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prewarm-s", type=float, default=0.4,
                    help="seconds of the same step, untimed, ahead of the warm-up steps: bench.py's protocol (a fresh box needs ~0.4 s of "
                         "load before its clocks settle; the few milliseconds of warm-up + timed steps here end before that).  0 = none")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--image-size", type=int, default=512)
    ap.add_argument("--cprofile", action="store_true", help="after the timing: 20 more steps under cProfile, top functions on stderr")
    ap.add_argument("--torus-div", type=float, default=1.0, help="1.0: SURVEY 8(d) config 3 as written (the bench headline); 1.5: the lighter batch of rounds 1-3")
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

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import run_reference_suite as rrs  # the iopath / imageio stubs the reference's package needs in this image

    rrs._stub_missing_packages()
    import pytorch3d_amd.shim as shim

    shim.install(ref_root, patch_python=False)
    from pytorch3d.renderer import FoVOrthographicCameras, MeshRasterizer, RasterizationSettings
    from pytorch3d.structures import Meshes

    if args.mode == "patched":
        shim.patch_reference_python()
    import _util as U
    from pytorch3d_amd import _lib

    d = torch.device("cuda:0")
    B, H, K = args.batch, args.image_size, 8
    blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
    verts, faces = U.hetero_batch(B, seed=0, torus_div=args.torus_div)
    mesh0 = Meshes(verts=[v.to(d) for v in verts], faces=[f.to(d) for f in faces])
    V = int(mesh0.verts_packed().shape[0])
    deform = torch.zeros((V, 3), device=d, requires_grad=True)
    cams = FoVOrthographicCameras(device=d)
    rs = RasterizationSettings(image_size=H, blur_radius=blur, faces_per_pixel=K, perspective_correct=True,
                               clip_barycentric_coords=True)
    rast = MeshRasterizer(cameras=cams, raster_settings=rs)
    gen = torch.Generator().manual_seed(231)
    g_z = torch.randn((B, H, H, K), generator=gen).to(d)
    g_b = torch.randn((B, H, H, K, 3), generator=gen).to(d)
    g_d = torch.randn((B, H, H, K), generator=gen).to(d)

    def step():
        deform.grad = None
        frag = rast(mesh0.offset_verts(deform))
        torch.autograd.backward([frag.zbuf, frag.bary_coords, frag.dists], [g_z, g_b, g_d])
        return frag

    t_pre, n_pre = time.perf_counter(), 0
    while args.prewarm_s > 0 and time.perf_counter() - t_pre < args.prewarm_s:
        frag = step()
        torch.cuda.synchronize()
        n_pre += 1
    for _ in range(args.warmup):
        frag = step()
    torch.cuda.synchronize()
    lib = _lib.load()
    lib.p3d_profile_reset()
    lib.p3d_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frag = step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.steps * 1e3
    lib.p3d_profile_enable(0)
    kern = {k: round(ms / n * (n / args.steps), 4) for k, (n, ms) in sorted(_lib.profile_snapshot().items())}
    from pytorch3d_amd import _C as ours_C

    out = {"mode": args.mode, "torus_div": args.torus_div, "cover_recalls_hit_miss": list(ours_C.COVER_RECALLS), "ms_per_step": wall, "Mpix_s": B * H * H / (wall * 1e-3) / 1e6, "steps": args.steps, "prewarm_s": args.prewarm_s, "prewarm_steps": n_pre,
           "our_kernels_ms_per_step": kern, "our_kernels_sum_ms": round(sum(kern.values()), 4),
           "covered": float((frag.pix_to_face[..., 0] >= 0).float().mean()),
           "grad_finite": bool(torch.isfinite(deform.grad).all()),
           "reference": ref_root if ref_root != stage else "oracle/_ref/reference_py (staged copy)"}
    if args.mode == "patched":
        out["patched_calls"] = {k: v for k, v in shim.PATCH_CALLS.items()}
    print(json.dumps(out))
    if args.cprofile:
        import cProfile
        import pstats

        pr = cProfile.Profile()
        torch.cuda.synchronize()
        pr.enable()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        pr.disable()
        st = pstats.Stats(pr, stream=sys.stderr)
        st.sort_stats("cumulative").print_stats(45)
        st.sort_stats("tottime").print_stats(25)


if __name__ == "__main__":
    main()
