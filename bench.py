#!/usr/bin/env python
"""bench.py -- rasterized Mpix/s (fwd+bwd) at 512^2, faces_per_pixel=8, on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: BASELINE.json configs[2], a batch of 64
heterogeneous meshes (1k-20k faces each, log-uniform; tori and icospheres, random rotation, pinhole view from 2.7),
512x512, K=8, SoftRas blur, perspective-correct + clipped barycentrics: `rasterize_meshes` forward (face gather + coarse
binning + fine rasterization) and backward (SoftRas gradient to the packed vertices) through the L2 mirror's autograd
Function, driven by fixed random upstream gradients for zbuf / bary / dists (the reference's gradient check,
tests/test_rasterize_meshes.py:563-571).  Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]                      weak scaling: every rank its own batch of 64
                                                                             (N > 1 without a launcher: spawns the N ranks itself)
    python bench.py --jobs 512 [--gpus N]                                    BASELINE configs[4]: 512 fixed jobs = 8 sub-batches
                                                                             of 64 (generator seeds 0..7), rank r runs sub-batches
                                                                             r, r+G, ...; one pass = K "steps" (K = 8/G per rank)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

The only collective is the final gather of the depth images to rank 0 (RCCL over xGMI), inside the timed region; its
time is also reported separately (`gather_ms`).  Rank 0 prints ONE JSON line.

Beside the headline number the line carries (rank 0, N = 1 only, all outside the timed region):
  roofline       dominant kernel: algorithmic bytes per launch / average launch duration (HIP events on the launch stream)
  cpu_baseline   the reference's own CPU kernels (oracle/_ref) on a seeded 8-mesh subset of the batch at full resolution,
                 and the reference's pure-Python `rasterize_meshes_python` on BASELINE configs[0], both on this box's cores
  other_configs  BASELINE configs[1] (the reference's cow, 256^2, K=8, coarse+fine forward) and configs[3] (1M points,
                 512^2, K=10, rasterizer + alpha compositor, fwd+bwd): wall and kernel milliseconds
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# SURVEY.md 8(d) config 3 literally: torus(r = 0.4 + 0.2 u, R = 1.0) unscaled (pytorch3d/utils/torus.py:24-73), ~58 % of the pixels
# covered.  Rounds 1-3 quoted the headline on the same generator with the tori scaled by 1/1.5 (~31 % covered): that batch is
# still timed, as the extra key `workload_torus_div_1.5`, never as `value`.
TORUS_DIV = 1.0
TORUS_SCALE = "tori r=0.4+0.2u, R=1.0 unscaled as SURVEY.md 8(d) config 3 defines them (tests/_util.py::hetero_batch(torus_div=1.0))"


def usable_cores():
    """Cores this process may really use: the affinity mask and the cgroup CPU quota, not os.cpu_count().  The round-5 GPU boxes
    show 256 cores and grant 16 (`/sys/fs/cgroup/cpu.max` = 1600000 100000): 256 OpenMP / intra-op threads on 16 cores spend their
    time spinning at barriers (the round-4 GPU suite: 1123 s with 256 threads, 122 s with 16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--image-size", type=int, default=512)
    ap.add_argument("--faces-per-pixel", type=int, default=8)
    ap.add_argument("--jobs", type=int, default=0, help="BASELINE configs[4]: a fixed number of jobs (multiple of --batch), strong scaling")
    ap.add_argument("--prewarm-s", type=float, default=0.4,
                    help="seconds of the same step, untimed, BEFORE the --warmup steps: a fresh box reaches its steady clocks and a warm "
                         "allocator only after some tenths of a second of work, and the driver's `--steps 20 --warmup 5` is 65 ms in all "
                         "(round 4: 6456 Mpix/s in that form against 6648 over 200 steps).  Reported in the line as `prewarm_s`; 0 turns it off")
    ap.add_argument("--gather-checksum", action="store_true",
                    help="after the timed region: sha256 of the gathered depth images on rank 0, in job order (tests compare an N-rank run "
                         "with the one-rank run of the same jobs bit for bit)")
    ap.add_argument("--dry-run", action="store_true",
                    help="multi-rank plumbing only: every rank reports (RANK, LOCAL_RANK, device) over gloo, rank 0 prints the mapping and what "
                         "is wrong with it; no RCCL, no kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-reference-device", action="store_true", help="skip the same-GPU leg on the reference's own device kernels (vs_reference_device)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the timing of the unmodified reference MeshRasterizer through the shim")
    ap.add_argument("--cpu-budget-s", type=float, default=110.0,
                    help="wall-clock bound for the CPU baseline legs (the Python reference takes ~31 s of it; the C++ kernels then "
                         "run as many meshes of the seeded 8-mesh subset as fit, smallest / largest / middle first, at least "
                         "two: ~10-30 s of CPU work each, the backward single-threaded)")
    return ap.parse_args()


def sub_batches_of_rank(jobs, batch, rank, world):
    """BASELINE configs[4]: `jobs` render jobs = jobs / batch sub-batches (generator seeds 0, 1, ...);
    rank r owns sub-batches r, r + world, ...  (8 sub-batches over 3 ranks: 3 + 3 + 2).  Raises when a rank would get none."""
    if jobs % batch:
        raise SystemExit("--jobs must be a multiple of --batch")
    n_sub = jobs // batch
    if n_sub < world:
        raise SystemExit(f"--jobs {jobs}: {n_sub} sub-batches for {world} ranks -- a rank would have nothing to run")
    return list(range(rank, n_sub, world))  # an uneven deal is fine: the ranks with one sub-batch more set the time


def build_batch(n_meshes, seed, device, torus_div=TORUS_DIV):
    import _util as U
    import pytorch3d_amd as p3d

    verts, faces = U.hetero_batch(n_meshes, seed=seed, torus_div=torus_div)
    nfaces = [int(f.shape[0]) for f in faces]
    meshes = p3d.PackedMeshes([v.to(device) for v in verts], [f.to(device) for f in faces])
    return meshes, verts, faces, nfaces


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1, after the timed region).  The only place that touches oracle/.
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(verts, faces, H, W, K, blur, budget_s):
    """(i) the reference's pure-Python rasterize_meshes_python on BASELINE configs[0] -- the path north_star names --
    and (ii) the reference's C++ CPU kernels (oracle/_ref: RasterizeMeshesNaiveCpu + RasterizeMeshesBackwardCpu) on a
    seeded subset of 8 meshes of the bench batch (meshes are independent, so the batch rate is the subset's),
    full resolution, forward (multi-threaded over image rows) + backward (single-threaded by construction,
    rasterize_meshes_cpu.cpp:412-529)."""
    from oracle import oracle as orc

    cores = usable_cores()
    torch.set_num_threads(cores)
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))  # the C oracle's OpenMP loops, should the "port" leg run
    t_start = time.perf_counter()
    py = python_reference_baseline(orc)
    ref = orc.ref_module()
    kind = "reference" if ref is not None else "port"
    subset = torch.randperm(len(faces), generator=torch.Generator().manual_seed(0)).tolist()[:8]
    # visiting order inside the subset: smallest, largest, then towards the middle -- whatever part of it the budget admits
    # spans the 1k-20k range instead of sitting on two mid-sized meshes
    by_size = sorted(subset, key=lambda j: faces[j].shape[0])
    order = []
    while by_size:
        order.append(by_size.pop(0))
        if by_size:
            order.append(by_size.pop(-1))
    gen = torch.Generator().manual_seed(231)
    done, px, t_fwd, t_bwd, nf = 0, 0, 0.0, 0.0, []
    for j in order:
        if done >= 2 and time.perf_counter() - t_start > budget_s:
            break
        v, f = verts[j], faces[j]
        fv = v[f].contiguous()
        F = fv.shape[0]
        first = torch.zeros(1, dtype=torch.int64)
        count = torch.tensor([F], dtype=torch.int64)
        nbr = torch.full((F,), -1, dtype=torch.int64)
        t0 = time.perf_counter()
        if ref is not None:
            out = ref._rasterize_meshes_naive(fv, first, count, nbr, (H, W), blur, K, True, True, False)
        else:
            out = orc.rasterize_meshes_naive(fv, first, count, nbr, (H, W), blur, K, True, True, False)
        t1 = time.perf_counter()
        g = [torch.randn(o.shape, generator=gen) for o in out[1:]]
        t2 = time.perf_counter()
        if ref is not None:
            ref.rasterize_meshes_backward(fv, out[0], g[0], g[1], g[2], True, True)
        else:
            orc.rasterize_meshes_backward(fv, out[0], g[0], g[1], g[2], True, True)
        t3 = time.perf_counter()
        t_fwd += t1 - t0
        t_bwd += t3 - t2
        px += H * W
        nf.append(F)
        done += 1
    dt = t_fwd + t_bwd
    return {
        "value": px / dt / 1e6,
        "unit": "Mpix/s",
        "cores": cores,
        "cores_visible": os.cpu_count(),
        "kind": kind,
        "sample": f"{done} of the seeded 8-mesh subset (randperm seed 0; visited smallest / largest / inwards) of the batch, faces {nf}, {H}x{W}, K={K}, "
                  f"naive fwd {t_fwd:.1f} s (multi-threaded over rows, {cores} threads) + bwd {t_bwd:.1f} s (single-threaded, "
                  "as the reference CPU path is); batch of 64 = this x 8",
        "python_reference": py,
    }


def python_reference_baseline(orc):
    """BASELINE configs[0] exactly: ico_sphere(2) (320 faces), view from 2.7, 64x64, K=1, blur 0,
    the reference's `rasterize_meshes_python` (renderer/mesh/rasterize_meshes.py:404-619) on CPU.  The reference's Python
    package exists on the GPU box only as the staged, git-ignored copy under oracle/_ref/reference_py."""
    stage = os.path.join(ROOT, "oracle", "_ref", "reference_py")
    for cand in (os.environ.get("P3D_REFERENCE_ROOT", "/root/reference"), stage):
        if os.path.isdir(os.path.join(cand, "pytorch3d", "renderer")):
            ref_root = cand
            break
    else:
        return {"value": None, "reason": "the reference's Python package is not on this machine (neither /root/reference nor "
                                         "oracle/_ref/reference_py staged by oracle/stage_reference.py)"}
    ref = orc.ref_module()
    if ref is None:
        return {"value": None, "reason": "oracle/_ref/p3d_ref_cpu.so not built: pytorch3d.renderer needs a _C module to import"}
    try:
        import importlib

        sys.modules["pytorch3d._C"] = ref
        if ref_root not in sys.path:
            sys.path.insert(0, ref_root)
        import pytorch3d

        pytorch3d._C = ref
        rm = importlib.import_module("pytorch3d.renderer.mesh.rasterize_meshes")
        from pytorch3d.structures import Meshes

        import _util as U

        v, f = U.ico_sphere(2)
        meshes = Meshes(verts=[U.to_ndc(v)], faces=[f])
        t0 = time.perf_counter()
        out = rm.rasterize_meshes_python(meshes, image_size=64, blur_radius=0.0, faces_per_pixel=1,
                                         perspective_correct=False, clip_barycentric_coords=False, cull_backfaces=False)
        dt = time.perf_counter() - t0
        # plumbing check of configs[0]: identical pix_to_face to the reference's C++ CPU rasterizer
        fv = meshes.verts_packed()[meshes.faces_packed()]
        cpp = ref._rasterize_meshes_naive(fv, meshes.mesh_to_faces_packed_first_idx(), meshes.num_faces_per_mesh(),
                                          torch.full((fv.shape[0],), -1, dtype=torch.int64), (64, 64), 0.0, 1, False, False, False)
        same = bool(torch.equal(out[0], cpp[0]))
        pairs = 64 * 64 * int(f.shape[0])
        return {"value": 64 * 64 / dt / 1e6, "unit": "Mpix/s", "seconds": dt, "pixel_face_pairs_per_s": pairs / dt, "cores": 1,
                "config": "BASELINE configs[0]: ico_sphere(2) 320 faces, 64x64, K=1, blur 0, rasterize_meshes_python (forward)",
                "pix_to_face_equals_cpp_cpu": same,
                "extrapolated_seconds_for_one_5k_face_512x512_mesh": 512 * 512 * 5000 / (pairs / dt)}
    except Exception as e:  # never sink the measurement
        return {"value": None, "reason": repr(e)}


# ---------------------------------------------------------------------------------------------------------------
# The other BASELINE configs (N = 1, after the timed region): extra keys, not the headline.
# ---------------------------------------------------------------------------------------------------------------
def _timed(lib, _lib, fn, iters=20, warm=3):
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
    return wall, {k: round(ms / n, 4) for k, (n, ms) in sorted(prof.items())}


def other_configs(lib, _lib, device):
    import numpy as np

    import pytorch3d_amd as p3d
    from pytorch3d_amd import _C

    out = {}
    # configs[1]: the reference's cow as MeshRasterizer.transform leaves it (tests/golden/cow_ref.npz), 256^2, K=8
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "cow_ref.npz"))
        m = p3d.PackedMeshes([torch.from_numpy(g["verts_ndc"]).to(device)], [torch.from_numpy(g["faces"]).long().to(device)])
        H = int(g["image_size"])
        K, blur = int(g["K"]), float(g["blur_radius"])

        def c2():
            return p3d.rasterize_meshes(m, image_size=H, blur_radius=blur, faces_per_pixel=K, perspective_correct=True,
                                        clip_barycentric_coords=True)

        wall, kern = _timed(lib, _lib, c2)
        F = int(g["faces"].shape[0])
        alg = H * H * K * 28 + F * 44 + 16
        out["config2_cow_256_k8_fwd"] = {"wall_ms": round(wall, 4), "kernels_ms": kern, "kernel_sum_ms": round(sum(kern.values()), 4),
                                         "algorithmic_bytes": alg, "gbps_of_wall": alg / (wall * 1e-3) / 1e9, "faces": F}
        # the same call (`_C.rasterize_meshes` on the gathered face vertices) replayed from a HIP graph: five launches, the gaps between
        # which are what separates wall from kernel sum on one small image (the C ABI allocates nothing and never synchronises)
        try:
            fv = m.verts_packed()[m.faces_packed()].contiguous()
            first = torch.zeros(1, dtype=torch.int64, device=device)
            cnt = torch.tensor([F], dtype=torch.int64, device=device)
            nbr = torch.full((F,), -1, dtype=torch.int64, device=device)
            cargs = (fv, first, cnt, nbr, (H, H), blur, K, 16, max(10000, F // 5), True, True, False)
            ref = _C.rasterize_meshes(*cargs)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    _C.rasterize_meshes(*cargs)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                gout = _C.rasterize_meshes(*cargs)

            def wall_of(fn, iters=200):
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / iters * 1e3

            eager_ms = wall_of(lambda: _C.rasterize_meshes(*cargs))
            replay_ms = wall_of(graph.replay)
            graph.replay()
            torch.cuda.synchronize()
            # the same operator through the two flavours of the boundary: the ctypes module (default) and the compiled pybind module
            # (pytorch3d_amd/csrc/bind.cpp) -- same five launches, different host side
            flav = {"ctypes_call_ms": round(eager_ms, 4)}
            try:
                from pytorch3d_amd import build_bind

                pyb = build_bind.load()
                same = all(torch.equal(a, b) for a, b in zip(pyb.rasterize_meshes(*cargs), ref))
                flav.update(pybind_call_ms=round(wall_of(lambda: pyb.rasterize_meshes(*cargs)), 4), identical_outputs=bool(same))
            except Exception as e:
                flav["pybind"] = "not available: %r" % (e,)
            out["config2_cow_256_k8_fwd"]["boundary_flavours"] = flav
            out["config2_cow_256_k8_fwd"]["hip_graph"] = {
                "eager_C_call_ms": round(eager_ms, 4), "graph_replay_ms": round(replay_ms, 4),
                "identical_outputs": bool(all(torch.equal(a, b) for a, b in zip(gout, ref))),
                "note": "_C.rasterize_meshes (binning + fine, 4-5 launches) captured with torch.cuda.graph and replayed"}
        except Exception as e:
            out["config2_cow_256_k8_fwd"]["hip_graph"] = {"error": repr(e)}
    except Exception as e:
        out["config2_cow_256_k8_fwd"] = {"error": repr(e)}
    # configs[3]: 1M points, 512^2, K=10, r=0.01, rasterizer + alpha compositor, forward + backward
    try:
        gen = torch.Generator().manual_seed(0)
        P, H, K, r, C = 1_000_000, 512, 10, 0.01, 3
        pts = torch.cat([torch.rand(P, 2, generator=gen) * 2 - 1, torch.rand(P, 1, generator=gen) * 2 + 0.5], 1).to(device)
        feats = torch.rand(C, P, generator=gen).to(device)
        first = torch.zeros(1, dtype=torch.int64, device=device)
        count = torch.full((1,), P, dtype=torch.int64, device=device)
        radius = torch.full((P,), r, device=device)
        gz = torch.randn((1, H, H, K), generator=gen).to(device)
        gd = torch.randn((1, H, H, K), generator=gen).to(device)
        gi = torch.randn((1, C, H, H), generator=gen).to(device)

        def c4():
            idx, zbuf, dists = _C.rasterize_points(pts, first, count, (H, H), radius, K, 32, 200000)
            alphas = (1 - dists / (r * r)).clamp(0, 1).permute(0, 3, 1, 2)
            pidx = idx.long().permute(0, 3, 1, 2)
            img = _C.accum_alphacomposite(feats, alphas, pidx)
            gf, ga = _C.accum_alphacomposite_backward(gi, feats, alphas, pidx)
            gp = _C.rasterize_points_backward(pts, idx, gz, gd)
            return img, gf, ga, gp

        wall, kern = _timed(lib, _lib, c4)
        # algorithmic bytes per launch: points fwd N*H*W*K*12 + P*16; bwd N*H*W*K*12 + P*12 read + P*12 written;
        # compositor fwd N*H*W*K*(8+4) + min(N*H*W*K, P)*C*4 + N*C*H*W*4; bwd the same reads + N*C*H*W*4 + N*K*H*W*4 + C*P*4
        px = H * H
        alg4 = {
            "points_fine": px * K * 12 + P * 16,
            "points_backward": px * K * 12 + P * 12 + P * 12,
            "alpha_composite_fwd": px * K * 12 + min(px * K, P) * C * 4 + C * px * 4,
            "alpha_composite_bwd": px * K * 12 + min(px * K, P) * C * 4 + C * px * 4 + C * px * 4 + px * K * 4 + C * P * 4,
        }
        per_kernel = {k: {"avg_ms": kern[k], "algorithmic_bytes": b, "algorithmic_gbps": b / (kern[k] * 1e-3) / 1e9,
                          "frac_of_peak": b / (kern[k] * 1e-3) / 1e9 / HBM_PEAK_GBPS} for k, b in alg4.items() if kern.get(k)}
        total = sum(alg4.values())
        out["config4_points_1m_512_k10_fwd_bwd"] = {"wall_ms": round(wall, 4), "kernels_ms": kern,
                                                    "kernel_sum_ms": round(sum(kern.values()), 4),
                                                    "algorithmic_bytes": total, "per_kernel": per_kernel,
                                                    "gbps_of_kernel_sum": total / (sum(kern.values()) * 1e-3) / 1e9,
                                                    "frac_of_peak_kernel_sum": total / (sum(kern.values()) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                                    "note": "wall includes the torch glue between the operators (alphas, permutes); 219 MB "
                                                            "on one image: latency / binning bound, not bandwidth bound (SURVEY 8(d))"}
        # the same chain as PointsRenderer writes it (weights = 1 - dists / r^2, no clamp; features (P, C); image (N, H, W, C)) on the
        # two fused launches of round 6 (include/p3d_amd.h: p3d_rasterize_points_composite / _composite_backward)
        feats_pc = feats.t().contiguous()
        gi_nhwc = gi.permute(0, 2, 3, 1).contiguous()
        inv = _C.inv_r2_of(r)

        def c4_fused():
            idx, zbuf, dists, img = _C.rasterize_points_composite(pts, first, count, (H, H), radius, feats_pc, inv, K, 32, 200000)
            gp, gf = _C.rasterize_points_composite_backward(pts, feats_pc, idx, dists, gi_nhwc, inv)
            return img, gf, gp

        wall_f, kern_f = _timed(lib, _lib, c4_fused)
        alg4f = {
            "points_fine": px * K * 12 + P * 16 + min(px * K, P) * C * 4 + px * C * 4,
            "points_composite_bwd": px * K * 8 + px * C * 4 + min(px * K, P) * (C * 4 + 8) + P * (3 + C) * 4,
        }
        per_kernel_f = {k: {"avg_ms": kern_f[k], "algorithmic_bytes": b, "algorithmic_gbps": b / (kern_f[k] * 1e-3) / 1e9,
                            "frac_of_peak": b / (kern_f[k] * 1e-3) / 1e9 / HBM_PEAK_GBPS} for k, b in alg4f.items() if kern_f.get(k)}
        # (same fragments either way; the operator chain above clamps its alphas, so the images are compared in the tests, not here)
        out["config4_points_1m_512_k10_fwd_bwd_fused"] = {
            "wall_ms": round(wall_f, 4), "kernels_ms": kern_f, "kernel_sum_ms": round(sum(kern_f.values()), 4),
            "algorithmic_bytes": sum(alg4f.values()), "per_kernel": per_kernel_f,
            "vs_operator_chain": {"wall": round(wall / wall_f, 3), "kernel_sum": round(sum(kern.values()) / sum(kern_f.values()), 3)},
            "note": "rasterizer + weights + alpha compositor forward as ONE chain of launches (compositing in the fine kernel's epilogue) and "
                    "their backward as one kernel; gradients to points (x, y) and features; bit-equal image, tests/test_gpu_render_points.py"}
    except Exception as e:
        out["config4_points_1m_512_k10_fwd_bwd"] = {"error": repr(e)}
    return out


def config5_one_gpu(args, device, H, W, K, blur, headline_mpix):
    """BASELINE configs[4] on ONE GPU, the N = 1 anchor of the scaling curve: 512 jobs = 8 sub-batches of
    64 (configs[2] generator, seeds 0..7) back to back through the headline's own step, then the job's one collective -- the gather
    of the depth images to rank 0 (`sharding.gather_batch`: with one rank the batch itself)."""
    import pytorch3d_amd as p3d
    from pytorch3d_amd import sharding

    B, n_sub = args.batch, 8
    gen = torch.Generator().manual_seed(231)
    g = [torch.randn(sh, generator=gen).to(device) for sh in ((B, H, W, K), (B, H, W, K, 3), (B, H, W, K))]
    subs = []
    for seed in range(n_sub):
        meshes, _, _, nfaces = build_batch(B, seed=seed, device=device)
        subs.append((meshes, meshes.verts_packed().clone().requires_grad_(True), sum(nfaces)))

    def one_pass():
        kept = []
        for meshes, vp, _ in subs:
            vp.grad = None
            p2f, zbuf, bary, dists = p3d.rasterize_meshes(meshes.update_verts_packed(vp), image_size=(H, W), blur_radius=blur,
                                                          faces_per_pixel=K, perspective_correct=True, clip_barycentric_coords=True)
            torch.autograd.backward([zbuf, bary, dists], g)
            kept.append(zbuf)
        shard = torch.cat([z[..., 0].detach() for z in kept], 0).contiguous()
        return sharding.gather_batch(shard, [B * n_sub], dst=0)

    one_pass()
    torch.cuda.synchronize()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        final = one_pass()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[1]
    mpix = n_sub * B * H * W / dt / 1e6
    return {"jobs": n_sub * B, "sub_batches": n_sub, "seconds_per_pass": dt, "ms_per_sub_batch": dt / n_sub * 1e3, "Mpix_s": mpix,
            "vs_headline": mpix / headline_mpix, "gathered_shape": list(final.shape), "total_faces": sum(s[2] for s in subs),
            "passes_timed": 3, "how": "median of 3 passes; the same as `python bench.py --jobs 512 --gpus 1`, inside the default line"}


def reference_device_leg(device, verts_cpu, faces_cpu, H, W, K, blur, our_ms_per_step):
    """The reference's OWN device kernels (pytorch3d/csrc/rasterize_meshes/*.cu as a ROCm build of pytorch3d compiles them:
    torch hipify + hipcc, default flags; oracle/build_ref_hip.py -> oracle/_ref/p3d_ref_hip.so) on the bench batch on this GPU:
    `_C.rasterize_meshes` (coarse + fine, bin_size / max_faces_per_bin as the reference's Python wrapper picks them,
    renderer/mesh/rasterize_meshes.py:201-222) + `_C.rasterize_meshes_backward`.  A checker leg, after the timed region."""
    from oracle import oracle as orc

    ref = orc.ref_hip_module(nofma=False)
    if ref is None:
        return {"value": None, "reason": "oracle/_ref/p3d_ref_hip.so is not built on this machine"}
    B = len(faces_cpu)
    fv = torch.cat([v[f] for v, f in zip(verts_cpu, faces_cpu)], 0).contiguous().to(device)
    counts = torch.tensor([int(f.shape[0]) for f in faces_cpu], dtype=torch.int64)
    first = (torch.cumsum(counts, 0) - counts).to(device)
    counts = counts.to(device)
    F = int(fv.shape[0])
    nbr = torch.full((F,), -1, dtype=torch.int64, device=device)
    M = int(max(10000, F / 5))
    fargs = (fv, first, counts, nbr, (H, W), blur, K, 32, M, True, True, False)
    gen = torch.Generator().manual_seed(231)
    g = [torch.randn(sh, generator=gen).to(device) for sh in ((B, H, W, K), (B, H, W, K, 3), (B, H, W, K))]

    def ms(fn, iters):
        ts = []
        for _ in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r = fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return sorted(ts)[len(ts) // 2], r

    out = ref.rasterize_meshes(*fargs)  # warm-up
    torch.cuda.synchronize()
    t_f, out = ms(lambda: ref.rasterize_meshes(*fargs), 2)
    ref.rasterize_meshes_backward(fv, out[0], g[0], g[1], g[2], True, True)
    t_b, _ = ms(lambda: ref.rasterize_meshes_backward(fv, out[0], g[0], g[1], g[2], True, True), 5)
    mpix = B * H * W / ((t_f + t_b) * 1e-3) / 1e6
    return {"value": mpix, "unit": "Mpix/s", "forward_ms": t_f, "backward_ms": t_b, "ours_ms_per_step": our_ms_per_step,
            "speedup": (t_f + t_b) / our_ms_per_step, "faces": F, "max_faces_per_bin": M,
            "kind": "the reference's rasterize_meshes.cu / rasterize_coarse.cu kernels, hipified and compiled for gfx950 as a checker "
                    "(oracle/build_ref_hip.py); same operator boundary, same inputs, torch events on the current stream"}


def config4_cpu_baseline(budget_s=40.0):
    """BASELINE.md B3 / B4 on this box's host cores (the reference's C++ CPU kernels, oracle/_ref/p3d_ref_cpu.so, and its Python
    interpolate_face_attributes_python): config 4 at the reduced size BASELINE.md measured (P = 100k, 128^2, K = 10, r = 0.01) with
    the factor to the full size stated; the compositor at its full size (it is cheap); a 64 x 64 x 8 fragment for the interpolation."""
    from oracle import oracle as orc

    ref = orc.ref_module()
    if ref is None:
        return {"value": None, "reason": "oracle/_ref/p3d_ref_cpu.so is not built on this machine"}
    cores = usable_cores()
    torch.set_num_threads(cores)
    gen = torch.Generator().manual_seed(0)
    P, H, K, r, C = 100_000, 128, 10, 0.01, 3
    pts = torch.cat([torch.rand(P, 2, generator=gen) * 2 - 1, torch.rand(P, 1, generator=gen) * 2 + 0.5], 1)
    first, count = torch.zeros(1, dtype=torch.int64), torch.full((1,), P, dtype=torch.int64)
    radius = torch.full((P,), r)
    t0 = time.perf_counter()
    idx, zbuf, dists = ref._rasterize_points_naive(pts, first, count, (H, H), radius, K)
    t_rf = time.perf_counter() - t0
    gz, gd = torch.randn(zbuf.shape, generator=gen), torch.randn(dists.shape, generator=gen)
    t0 = time.perf_counter()
    ref.rasterize_points_backward(pts, idx, gz, gd)
    t_rb = time.perf_counter() - t0
    feats = torch.rand(C, P, generator=gen)
    alphas = (1 - dists / (r * r)).clamp(0, 1).permute(0, 3, 1, 2).contiguous()
    pidx = idx.long().permute(0, 3, 1, 2).contiguous()
    t0 = time.perf_counter()
    img = ref.accum_alphacomposite(feats, alphas, pidx)
    t_cf = time.perf_counter() - t0
    gi = torch.randn(img.shape, generator=gen)
    t0 = time.perf_counter()
    ref.accum_alphacomposite_backward(gi, feats, alphas, pidx)
    t_cb = time.perf_counter() - t0
    total = t_rf + t_rb + t_cf + t_cb
    pairs_factor = (1_000_000 * 512 * 512) / (P * H * H)  # the naive rasterizer tests every (pixel, point) pair
    pixel_factor = (512 * 512) / (H * H)                  # backward and compositor work per pixel slot
    full_s = t_rf * pairs_factor + (t_rb + t_cf + t_cb) * pixel_factor
    out = {"kind": "reference", "cores": cores, "unit": "Mpix/s",
           "value": H * H / total / 1e6,
           "sample": f"P={P}, {H}x{H}, K={K}, r={r}: RasterizePointsNaiveCpu fwd {t_rf:.2f} s + RasterizePointsBackwardCpu {t_rb * 1e3:.1f} ms + "
                     f"alphaCompositeCpuForward {t_cf * 1e3:.1f} ms + Backward {t_cb * 1e3:.1f} ms (rasterize_points_cpu.cpp:14-96, "
                     "alpha_composite_cpu.cpp:17-124; single-threaded as the reference writes them)",
           "hits_per_pixel": float((idx >= 0).float().sum() / (H * H)),
           "scale_to_full_config4": {"pixel_point_pairs": pairs_factor, "pixel_slots": pixel_factor, "extrapolated_seconds": full_s,
                                     "extrapolated_Mpix_s": 512 * 512 / full_s / 1e6}}
    # B4: the reference has no C++ CPU interpolation (interp_face_attrs.h:29-35); its Python formulation on a 64 x 64 x 8 fragment
    try:
        N, Hf, Kf, F, D = 1, 64, 8, 5000, 3
        p2f = torch.randint(-1, F, (N, Hf, Hf, Kf), generator=gen)
        bary = torch.rand(N, Hf, Hf, Kf, 3, generator=gen)
        attrs = torch.rand(F, 3, D, generator=gen)
        fn, what = None, None
        try:  # the reference's own function, when its Python package is on this machine (python_reference_baseline staged the import)
            from pytorch3d.ops.interp_face_attrs import interpolate_face_attributes_python as fn

            what = "the reference's interpolate_face_attributes_python (ops/interp_face_attrs.py:86-102) on CPU tensors"
        except Exception:
            def fn(p2f, bary, attrs):  # the same formulation restated: gather + weighted sum + mask
                mask = p2f < 0
                idxe = p2f.clone().view(-1, 1, 1).expand(-1, 3, attrs.shape[-1])
                idxe = torch.where(idxe < 0, torch.zeros_like(idxe), idxe)
                pix = attrs.gather(0, idxe).view(tuple(p2f.shape) + (3, attrs.shape[-1]))
                vals = (bary[..., None] * pix).sum(dim=-2)
                vals[mask] = 0
                return vals

            what = "the torch formulation of ops/interp_face_attrs.py:86-102 restated (the reference's package is not importable here)"
        fn(p2f, bary, attrs)
        t0 = time.perf_counter()
        fn(p2f, bary, attrs)
        t_i = time.perf_counter() - t0
        out["interp_face_attrs_python"] = {"seconds": t_i, "fragment": [N, Hf, Hf, Kf], "D": D, "Mpix_s": Hf * Hf / t_i / 1e6, "cores": cores,
                                           "what": what}
    except Exception as e:
        out["interp_face_attrs_python"] = {"value": None, "reason": repr(e)}
    return out


def dry_run(world, rank, local_rank):
    """`--dry-run`: the launch plumbing of an N-GPU run without RCCL and without a kernel -- what a first contact with an 8-GPU node
    should not have to debug.  Every rank reports (RANK, LOCAL_RANK, the device it would bind, visible devices, pid) over gloo;
    rank 0 checks that the ranks are 0..N-1, that LOCAL_RANK -> device is one-to-one and inside the visible devices, and prints
    ONE JSON line.  Exit code 0 also when the mapping has problems on this box (they are in the line: a one-GPU box running
    `--gpus 8 --dry-run` lists seven devices that do not exist)."""
    import torch.distributed as dist

    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    mine = {"rank": rank, "local_rank": local_rank, "device": f"cuda:{local_rank}", "devices_visible": visible, "pid": os.getpid(),
            "master": f"{os.environ.get('MASTER_ADDR', '?')}:{os.environ.get('MASTER_PORT', '?')}",
            "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    rows = [mine]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        rows = [None] * world
        dist.all_gather_object(rows, mine)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        problems = []
        if sorted(r["rank"] for r in rows) != list(range(world)):
            problems.append("ranks are not 0..N-1: %r" % [r["rank"] for r in rows])
        if len({r["local_rank"] for r in rows}) != world:
            problems.append("LOCAL_RANK is not one-to-one: %r" % [r["local_rank"] for r in rows])
        for r in rows:
            if r["local_rank"] >= r["devices_visible"]:
                problems.append(f"rank {r['rank']}: LOCAL_RANK {r['local_rank']} has no device (visible: {r['devices_visible']})")
        if world > 1 and any(r["ipc_mode_legacy"] != "0" for r in rows):
            problems.append("HSA_ENABLE_IPC_MODE_LEGACY is not 0 on every rank: RCCL across processes needs dmabuf IPC on this driver")
        print(json.dumps({"dry_run": True, "n_gpus": world, "ok": not problems, "problems": problems, "mapping": rows,
                          "backend_planned": "nccl (RCCL over xGMI), gloo if it cannot be brought up"}), flush=True)
    return 0


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-execute this script under torch.distributed.run with one rank per
    GPU (what the driver does itself for N > 1, and what the reference's only multi-device test does with
    nn.DataParallel, tests/test_render_multigpu.py:127-184).  The children print the ONE JSON line (rank 0)."""
    import socket
    import subprocess

    n = args.gpus
    have = torch.cuda.device_count()
    if have < n and not (args.dry_run or os.environ.get("P3D_BENCH_TEST_BACKEND") or os.environ.get("P3D_BENCH_SHARED_GPU")):
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
    return subprocess.call(cmd, env=env)


def _child_env():
    """Environment of the drop-in subprocesses: torch's intra-op / OpenMP pools sized to the cores the box grants (usable_cores)."""
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(usable_cores()))
    env.setdefault("MKL_NUM_THREADS", str(usable_cores()))
    return env


def dropin_timing(batch, image_size):
    """The UNMODIFIED reference `MeshRasterizer.forward` + backward through the shim on the same batch, `_C` only and with
    shim.install(patch_python=True) (profiles/dropin_timing.py, one subprocess each: the shim must not leak into this
    process).  -> {"c_only": {...}, "patched": {...}} or a reason."""
    import subprocess

    out = {}
    for mode, div in (("c_only", TORUS_DIV), ("patched", TORUS_DIV)):
        key = mode if div == TORUS_DIV else f"{mode}_torus_div_{div}"
        try:
            res = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "dropin_timing.py"), "--mode", mode, "--batch", str(batch),
                                  "--image-size", str(image_size), "--torus-div", str(div)], capture_output=True, text=True, timeout=600, cwd=ROOT,
                                 env=_child_env())
            lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
            out[key] = json.loads(lines[-1]) if lines else {"value": None, "reason": (res.stderr or res.stdout)[-400:]}
        except Exception as e:  # never sink the measurement
            out[key] = {"value": None, "reason": repr(e)}
    return out


def dropin_points_timing():
    """SURVEY.md 8(d) config 4 as written: the UNMODIFIED reference PointsRenderer(PointsRasterizer, AlphaCompositor) on 1M points,
    512^2, K = 10, loss = sum(image * g), autograd backward to points and features, through the shim -- `_C` only and with
    shim.install(patch_python=True) (profiles/dropin_points_timing.py, one subprocess each)."""
    import subprocess

    out = {}
    for mode in ("c_only", "patched", "patched_operators"):  # patched: the fused PointsRenderer node; _operators: with it switched off
        try:
            flags = ["--mode", "patched", "--no-fuse"] if mode == "patched_operators" else ["--mode", mode]
            res = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "dropin_points_timing.py")] + flags, capture_output=True,
                                 text=True, timeout=300, cwd=ROOT, env=_child_env())
            lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
            out[mode] = json.loads(lines[-1]) if lines else {"value": None, "reason": (res.stderr or res.stdout)[-400:]}
        except Exception as e:  # never sink the measurement
            out[mode] = {"value": None, "reason": repr(e)}
    return out


def light_workload_sensitivity(device, B, H, W, K, blur, steps=20):
    """The same step on the generator's tori scaled by 1 / 1.5 (the batch rounds 1-3 quoted: a third of the pixels covered
    instead of 58 %): Mpix/s counts N*H*W whatever the coverage, so the lighter batch reads higher.  Extra key only."""
    import pytorch3d_amd as p3d

    meshes, _, _, nfaces = build_batch(B, seed=0, device=device, torus_div=1.5)
    vp = meshes.verts_packed().clone().requires_grad_(True)
    gen = torch.Generator().manual_seed(231)
    g = [torch.randn(s, generator=gen).to(device) for s in ((B, H, W, K), (B, H, W, K, 3), (B, H, W, K))]

    def step():
        vp.grad = None
        p2f, zbuf, bary, dists = p3d.rasterize_meshes(meshes.update_verts_packed(vp), image_size=(H, W), blur_radius=blur, faces_per_pixel=K,
                                                      perspective_correct=True, clip_barycentric_coords=True)
        torch.autograd.backward([zbuf, bary, dists], g)
        return p2f

    from pytorch3d_amd import _lib

    lib = _lib.load()
    for _ in range(3):
        p2f = step()
    torch.cuda.synchronize()
    lib.p3d_profile_reset()
    lib.p3d_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        p2f = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    lib.p3d_profile_enable(0)
    prof = _lib.profile_snapshot()
    return {"ms_per_step": ms, "Mpix_s": B * H * W / (ms * 1e-3) / 1e6, "covered_pixel_fraction": float((p2f[..., 0] >= 0).float().mean()),
            "pixel_slot_fill": float((p2f >= 0).float().mean()), "total_faces": sum(nfaces), "steps": steps,
            "kernels_ms": {k: round(t / n, 4) for k, (n, t) in sorted(prof.items())},
            "note": "hetero_batch(torus_div=1.5): the lighter batch of rounds 1-3; not the headline workload"}


def main():
    args = parse()
    torch.set_num_threads(usable_cores())  # host-side glue only (seeded randn of the upstream gradients); see usable_cores()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is used", file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        raise SystemExit(dry_run(world, rank, local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X); no CPU fallback exists for the product path")
    # P3D_BENCH_TEST_BACKEND=gloo maps every rank to cuda:0 and uses gloo: lets the multi-rank code path (barriers,
    # max-over-ranks timing, final gather) be exercised on a single-GPU box.  Never set by the driver.
    test_backend = os.environ.get("P3D_BENCH_TEST_BACKEND")
    # P3D_BENCH_SHARED_GPU=1 (tests only, with P3D_BENCH_FAIL_NCCL=1): every rank on cuda:0 but the REAL backend choice, so that
    # a single-GPU box exercises the "nccl failed -> gloo" fallback below.  (RCCL itself HANGS with two ranks on one device --
    # measured in round 4, 900 s -- so the failure is injected, not provoked.)
    if test_backend or os.environ.get("P3D_BENCH_SHARED_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist_on = world > 1
    backend, backend_note = None, None
    if dist_on:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = test_backend or "nccl"
        try:
            if test_backend:
                dist.init_process_group(test_backend, rank=rank, world_size=world)
            else:
                if os.environ.get("P3D_BENCH_FAIL_NCCL"):  # tests only: stands in for an RCCL that cannot be brought up
                    raise RuntimeError("P3D_BENCH_FAIL_NCCL is set")
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
                probe = torch.ones(1, device=device)
                dist.all_reduce(probe)  # RCCL opens its communicator lazily: fail HERE, where every rank can still fall back alike
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError(f"all_reduce over RCCL returned {probe.item()} for {world} ranks")
        except Exception as e:  # RCCL unusable on this node (IPC mode, xGMI topology ...): keep the measurement, say so in the line
            backend_note = f"nccl failed ({type(e).__name__}: {str(e)[:300]}); barriers, timing and the final gather run over gloo (host memory)"
            print(f"[bench rank {rank}] {backend_note}", file=sys.stderr, flush=True)
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:
                pass
            # an explicit store on the next port, hosted by rank 0: under torchrun an `init_method` URL makes every rank a CLIENT of
            # the agent's store (TORCHELASTIC_USE_AGENT_STORE), which nobody hosts on a second port -- measured: it hangs
            import datetime

            port = int(os.environ.get("MASTER_PORT", "29500")) + 1
            store = dist.TCPStore(os.environ["MASTER_ADDR"], port, world, is_master=(rank == 0), timeout=datetime.timedelta(seconds=120))
            dist.init_process_group("gloo", store=store, rank=rank, world_size=world)
            backend = "gloo"
    comm_dev = device if backend == "nccl" else torch.device("cpu")  # where tensors of a collective have to live

    import pytorch3d_amd as p3d
    from pytorch3d_amd import _lib
    from pytorch3d_amd import sharding

    lib = _lib.load()
    H = W = args.image_size
    K = args.faces_per_pixel
    sigma = 1e-4
    blur = math.log(1.0 / 1e-4 - 1.0) * sigma  # SoftRas convention, tests/test_render_meshes.py:462
    B = args.batch

    # ---- which sub-batches this rank owns ---------------------------------------------------------------------
    jobs_mode = args.jobs > 0
    if jobs_mode:
        seeds = sub_batches_of_rank(args.jobs, B, rank, world)  # sub-batch s = generator seed s
        steps = len(seeds)
    else:
        seeds = [rank]
        steps = args.steps
    batches = []
    for s in seeds:
        meshes, verts_cpu, faces_cpu, nfaces = build_batch(B, seed=s, device=device)
        vp = meshes.verts_packed().clone().requires_grad_(True)
        batches.append((meshes, vp, sum(nfaces), verts_cpu, faces_cpu))
    gen = torch.Generator().manual_seed(231 + rank)
    g_z = torch.randn((B, H, W, K), generator=gen).to(device)
    g_b = torch.randn((B, H, W, K, 3), generator=gen).to(device)
    g_d = torch.randn((B, H, W, K), generator=gen).to(device)

    def step(i):
        meshes, vp, _, _, _ = batches[i % len(batches)]
        vp.grad = None
        m = meshes.update_verts_packed(vp)
        p2f, zbuf, bary, dists = p3d.rasterize_meshes(m, image_size=(H, W), blur_radius=blur, faces_per_pixel=K,
                                                      perspective_correct=True, clip_barycentric_coords=True)
        torch.autograd.backward([zbuf, bary, dists], [g_z, g_b, g_d])
        return p2f, zbuf

    own = [B] * world if not jobs_mode else [B * len(sub_batches_of_rank(args.jobs, B, r, world)) for r in range(world)]

    gather_state = {"ok": True, "how": "gather to rank 0", "error": None}

    def final_gather(depths):
        # the one collective of the job: the final depth images of every rank are gathered on rank 0
        shard = torch.cat([z[..., 0].detach() for z in depths], 0).contiguous().to(comm_dev)
        try:
            return sharding.gather_batch(shard, own, dst=0)
        except (RuntimeError, NotImplementedError) as e:  # a backend without gather: every rank raises alike
            gather_state.update(how="all_gather (gather raised: %s)" % str(e)[:120])
            try:
                return sharding.gather_batch(shard, own)
            except Exception as e2:  # reported, never fatal: the per-rank numbers are still worth having
                gather_state.update(ok=False, how="none", error=f"{type(e2).__name__}: {str(e2)[:300]}")
                return None

    # The driver's protocol to the letter first (W warm-up steps, K timed, nothing ahead of them): reported beside the headline as
    # `without_prewarm`, so that the two protocols never have to be compared across runs.
    cold = None
    if args.prewarm_s > 0 and not jobs_mode:
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        tc0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        cold_s = time.perf_counter() - tc0
        cold = {"ms_per_step": cold_s / steps * 1e3, "value": world * B * H * W * steps / cold_s / 1e6, "steps": steps, "warmup": args.warmup,
                "what": "this rank's own clock over the same K steps after W warm-up steps on a process that had run nothing before"}
    prewarm_steps = 0
    if args.prewarm_s > 0:
        # untimed, ahead of the W warm-up steps (see --prewarm-s): never inside the timed region, never counted as steps
        t_pw = time.perf_counter()
        while time.perf_counter() - t_pw < args.prewarm_s:
            for i in range(8):
                p2f, zbuf = step(i)
            torch.cuda.synchronize()
            prewarm_steps += 8
    for i in range(args.warmup):
        p2f, zbuf = step(i)
    if dist_on and args.warmup > 0:
        # part of the warmup: RCCL opens its point-to-point xGMI channels on the first gather (lazily, ~100 ms)
        del_me = final_gather([zbuf] * (steps if jobs_mode else 1))
        del del_me
    torch.cuda.synchronize()
    lib.p3d_profile_reset()
    lib.p3d_profile_enable(1)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kept = []
    ev[0].record()
    for i in range(steps):
        p2f, zbuf = step(i)
        ev[i + 1].record()
        if jobs_mode:
            kept.append(zbuf)
    tg0 = time.perf_counter()
    gather_ms = 0.0
    final = None
    if dist_on or args.gather_checksum:
        torch.cuda.synchronize()
        tg0 = time.perf_counter()
        final = final_gather(kept if jobs_mode else [zbuf])
        if not args.gather_checksum:
            final = None
    torch.cuda.synchronize()
    if dist_on:
        gather_ms = (time.perf_counter() - tg0) * 1e3
        dist.barrier()
    t1 = time.perf_counter()
    lib.p3d_profile_enable(0)
    prof = _lib.profile_snapshot()
    mine = torch.tensor([t1 - t0, gather_ms, (tg0 - t0) * 1e3 / max(steps, 1),
                         prof.get("mesh_fine", (1, 0.0))[1] / max(prof.get("mesh_fine", (1, 0.0))[0], 1),
                         prof.get("mesh_backward", (1, 0.0))[1] / max(prof.get("mesh_backward", (1, 0.0))[0], 1)],
                        dtype=torch.float64, device=comm_dev)
    per_rank = None
    if dist_on:
        # every rank's own clock: [seconds incl. gather, gather ms, compute ms per step, mesh_fine ms, mesh_backward ms]
        rows = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
        per_rank = [{"rank": r, "seconds": float(x[0]), "gather_ms": float(x[1]), "compute_ms_per_step": float(x[2]),
                     "mesh_fine_ms": float(x[3]), "mesh_backward_ms": float(x[4])} for r, x in enumerate(rows)]
        elapsed = max(p["seconds"] for p in per_rank)  # MAX over ranks, as the contract asks
        gather_ms = max(p["gather_ms"] for p in per_rank)
    else:
        elapsed, gather_ms = float(mine[0]), float(mine[1])
    per_step = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    median_ms = per_step[len(per_step) // 2]
    valid = p2f >= 0
    hit_frac = float(valid.float().mean().item())
    covered = int(valid.any(-1).sum().item())  # pixels of the last step's batch with at least one face
    # 16-pixel row segments with a face: what the backward walks when the forward hands it the row cover
    seg_w = (W + 15) // 16
    pad = torch.zeros((B, H, seg_w * 16), dtype=torch.bool, device=device)
    pad[:, :, :W] = valid.any(-1)
    cover_segments = int(pad.view(B, H, seg_w, 16).any(-1).sum().item())
    del pad
    total_faces = batches[(steps - 1) % len(batches)][2]

    gathered = None
    if rank == 0 and final is not None:
        import hashlib

        if jobs_mode:  # rank-major (rank r holds sub-batches r, r + world, ...) -> job order
            deal = [sb for r in range(world) for sb in sub_batches_of_rank(args.jobs, B, r, world)]
            pos = torch.tensor([deal.index(sb) for sb in range(len(deal))], device=final.device)
            final = final.view(len(deal), B, H, W)[pos].reshape(-1, H, W)
        gathered = {"shape": list(final.shape), "sha256": hashlib.sha256(final.detach().cpu().contiguous().numpy().tobytes()).hexdigest(),
                    "order": "job order (sub-batch = generator seed, ascending)" if jobs_mode else "rank order (rank r ran generator seed r)"}
    del final
    if rank == 0:
        pixels = args.jobs * H * W if jobs_mode else world * B * H * W * steps
        value = pixels / elapsed / 1e6
        # ---- roofline of the dominant kernel (algorithmic bytes) -----------------------------------
        px = B * H * W
        alg = {
            "mesh_fine": px * K * 28 + total_faces * 44 + 16 * B,
            "mesh_naive": px * K * 28 + total_faces * 44 + 16 * B,
            # SURVEY 8d's figure (every sample's gradients read) ...
            "mesh_backward": px * K * 28 + 2 * total_faces * 36,
        }
        # ... and what the backward has to move given the data: pix_to_face of every sample, gradient rows only of
        # pixels that hold a face (background rows carry no information and are skipped)
        compulsory = dict(alg)
        compulsory["mesh_backward"] = px * K * 8 + covered * K * 20 + 2 * total_faces * 36
        # ... and what it moves since round 3, when the forward's row cover tells it which 16-pixel row segments hold a
        # face at all (pix_to_face of the others is never read): reported beside the two above, never instead of them
        with_cover = {"mesh_backward": cover_segments * 16 * K * 8 + covered * K * 20 + 2 * total_faces * 36}
        kernels = {k: {"launches": n, "avg_ms": ms / n} for k, (n, ms) in prof.items()}
        dom = max((k for k in kernels if k in alg), key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches"])
        achieved = alg[dom] / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
        traffic, traffic_source, tj = None, None, {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # counters of another workload say nothing about this one: the file names the generator scale it was collected on
                if float(tj.get("_torus_div", 1.5)) == float(TORUS_DIV) and not jobs_mode and (B, H, W, K) == (64, 512, 512, 8):
                    traffic = tj.get(dom)
                    traffic_source = tj.get("_source", "profiles/traffic.json") + " (rocprofv3 PMC passes of an earlier run of this command; not measured in this run)"
                else:
                    traffic_source = "profiles/traffic.json was collected on another workload (torus_div %s): not used" % tj.get("_torus_div", 1.5)
            except Exception:
                traffic = None
        # "secondary, reported alongside: fp32 VALU issue rate and LDS bandwidth for the fine stage".  The counters
        # (SQ_INSTS_VALU per launch, from the same committed rocprofv3 PMC passes as `traffic`) priced at one wave64 instruction
        # per 4 cycles and SIMD: the time the launch cannot go under while it issues that many vector instructions.
        valu, lds_note = None, None
        try:
            props = torch.cuda.get_device_properties(device)
            simds = props.multi_processor_count * 4
            clock_ghz = 2.4  # MI355X peak engine clock, /opt/skills/guides/MI355X_MICROARCH.md (the sustained clock is lower: the bound is a floor)
            iss = (tj.get("_issue") or {}).get(dom) if traffic is not None else None
            if iss and iss.get("SQ_INSTS_VALU"):
                insts = float(iss["SQ_INSTS_VALU"])
                bound_ms = insts * 4.0 / simds / (clock_ghz * 1e9) * 1e3
                valu = {"insts_per_launch": insts, "cycles_per_inst": 4, "simds": simds, "clock_ghz": clock_ghz, "issue_bound_ms": bound_ms,
                        "frac": bound_ms / kernels[dom]["avg_ms"], "salu_insts_per_launch": iss.get("SQ_INSTS_SALU"),
                        "per_wave": insts / iss["SQ_WAVES"] if iss.get("SQ_WAVES") else None,
                        "source": traffic_source}
                if iss.get("SQ_LDS_IDX_ACTIVE"):
                    lds_note = {"lds_insts_per_launch": iss.get("SQ_INSTS_LDS"), "lds_active_cycles": iss.get("SQ_LDS_IDX_ACTIVE"),
                                "bank_conflict_cycles": iss.get("SQ_LDS_BANK_CONFLICT"),
                                "bank_conflict_frac": iss.get("SQ_LDS_BANK_CONFLICT", 0.0) / iss["SQ_LDS_IDX_ACTIVE"]}
        except Exception:
            valu = None
        hbm_frac = achieved / HBM_PEAK_GBPS
        roofline = {
            # the binding resource: whichever fraction is higher (both headline kernels sit on the vector issue
            # ceiling with HBM two thirds idle).  achieved / peak / frac stay the HBM figures the metric asks for.
            "bound": "valu" if (valu is not None and valu["frac"] > hbm_frac) else "hbm",
            "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": hbm_frac, "valu": valu, "lds": lds_note, "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": kernels[dom]["avg_ms"],
            "per_kernel": {k: {"avg_ms": round(kernels[k]["avg_ms"], 4), "algorithmic_gbps": alg[k] / (kernels[k]["avg_ms"] * 1e-3) / 1e9,
                               "compulsory_bytes": compulsory[k],
                               "compulsory_gbps": compulsory[k] / (kernels[k]["avg_ms"] * 1e-3) / 1e9,
                               "frac_of_peak_compulsory": compulsory[k] / (kernels[k]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                               **({"bytes_with_row_cover": with_cover[k],
                                   "frac_of_peak_with_row_cover": with_cover[k] / (kernels[k]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS}
                                  if k in with_cover else {})}
                           for k in kernels if k in alg},
            "step_algorithmic_gbps": (alg["mesh_fine"] + alg["mesh_backward"]) / (median_ms * 1e-3) / 1e9,
        }
        workload = (f"BASELINE configs[2]: batch of {B} heterogeneous meshes per GPU (1k-20k faces log-uniform, tori/icospheres; "
                    f"{TORUS_SCALE}), 512x512, faces_per_pixel=8, SoftRas blur, perspective-correct + clipped bary, fwd+bwd")
        if jobs_mode:
            workload = (f"BASELINE configs[4]: {args.jobs} jobs = {args.jobs // B} sub-batches of {B} (configs[2] generator, seeds "
                        f"0..{args.jobs // B - 1}), each rank runs {steps} sub-batches back to back; " + workload)
        out = {
            "metric": "rasterized Mpix/s (fwd+bwd) at 512^2 faces_per_pixel=8",
            "value": value,
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "prewarm_s": args.prewarm_s, "prewarm_steps": prewarm_steps, "without_prewarm": cold,
            "ms_per_step": elapsed / steps * 1e3,
            "ms_per_step_median": median_ms,
            "gather_ms": gather_ms,
            "gather": dict(gather_state, backend=backend, backend_note=backend_note, gathered=gathered) if (dist_on or gathered) else None,
            "per_rank": per_rank,
            "higher_is_better": True,
            "scaling": "strong" if jobs_mode else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "global_batch": args.jobs if jobs_mode else world * B, "image_size": [H, W], "faces_per_pixel": K, "blur_radius": blur,
                "total_faces_per_rank": total_faces, "pixel_slot_fill": hit_frac, "covered_pixel_fraction": covered / px,
                "parallelism": f"batch-sharded x{world}, final gather to rank 0 only",
                "path": "pytorch3d_amd.rasterize_meshes -- this package's L2 mirror of renderer/mesh/rasterize_meshes.py (autograd Function over "
                        "the C ABI); the unmodified reference MeshRasterizer over pytorch3d._C on the same batch is timed beside it: dropin_ms_per_step",
            },
            "roofline": roofline,
            "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in sorted(kernels.items())},
        }
        if world == 1 and not args.no_other_configs:
            try:
                out["other_configs"] = other_configs(lib, _lib, device)
            except Exception as e:
                out["other_configs"] = {"error": repr(e)}
        if world == 1 and not jobs_mode and not args.no_other_configs:
            try:
                out["workload_torus_div_1.5"] = light_workload_sensitivity(device, B, H, W, K, blur)
            except Exception as e:
                out["workload_torus_div_1.5"] = {"error": repr(e)}
        if world == 1 and not jobs_mode and not args.no_dropin:
            out["dropin"] = dropin_timing(B, H)
            out["dropin"]["mirror_ms_per_step"] = elapsed / steps * 1e3
            # what `from pytorch3d.renderer import MeshRasterizer` gets on the same batch, next to the headline's own path
            out["config"]["dropin_ms_per_step"] = {k: round(v["ms_per_step"], 4) for k, v in out["dropin"].items()
                                                   if isinstance(v, dict) and v.get("ms_per_step")}
            if B == 64 and H == 512 and isinstance(out.get("other_configs"), dict):
                out["other_configs"]["config4_points_renderer_dropin"] = dropin_points_timing()
        if world == 1 and not jobs_mode and not args.no_other_configs and B == 64 and H == 512 and isinstance(out.get("other_configs"), dict):
            try:  # the N = 1 anchor of BASELINE configs[4]'s scaling curve, timed on this build in this run
                out["other_configs"]["config5_jobs512_1gpu"] = config5_one_gpu(args, device, H, W, K, blur, value)
            except Exception as e:
                out["other_configs"]["config5_jobs512_1gpu"] = {"error": repr(e)}
        if world == 1 and not jobs_mode and not args.no_reference_device:
            try:  # the same-GPU baseline: the reference's own device kernels on the bench batch
                _, _, _, verts_cpu, faces_cpu = batches[0]
                rd = reference_device_leg(device, verts_cpu, faces_cpu, H, W, K, blur, elapsed / steps * 1e3)
                out["vs_reference_device"] = rd
                if rd.get("value"):
                    out["vs_baseline"] = value / rd["value"]
                    out["vs_baseline_note"] = ("BASELINE.md publishes no number for this metric; the ratio is against the reference's own device kernels "
                                               "compiled for this GPU and timed in this run (vs_reference_device)")
            except Exception as e:
                out["vs_reference_device"] = {"value": None, "error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                _, _, _, verts_cpu, faces_cpu = batches[0]
                out["cpu_baseline"] = cpu_baseline(verts_cpu, faces_cpu, H, W, K, blur, args.cpu_budget_s)
            except Exception as e:  # the baseline must never sink the GPU measurement
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
            if isinstance(out.get("other_configs"), dict) and isinstance(out["other_configs"].get("config4_points_1m_512_k10_fwd_bwd"), dict):
                try:
                    out["other_configs"]["config4_points_1m_512_k10_fwd_bwd"]["cpu_baseline"] = config4_cpu_baseline()
                except Exception as e:
                    out["other_configs"]["config4_points_1m_512_k10_fwd_bwd"]["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
