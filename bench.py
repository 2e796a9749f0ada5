#!/usr/bin/env python
"""bench.py -- rasterized Mpix/s (fwd+bwd) at 512^2, faces_per_pixel=8, on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: BASELINE.json configs[2],
a batch of 64 heterogeneous meshes (1k-20k faces each, log-uniform; tori and icospheres, random
rotation, pinhole view from 2.7), 512x512, K=8, SoftRas blur, perspective-correct + clipped
barycentrics: `rasterize_meshes` forward (face gather + coarse binning + fine rasterization) and
backward (SoftRas gradient to the packed vertices) through the L2 mirror's autograd Function,
driven by fixed random upstream gradients for zbuf / bary / dists (the reference's gradient
check, tests/test_rasterize_meshes.py:563-571).  Inputs are resident in HBM before the timed
region.  With --gpus N every rank rasterizes its own batch of 64 (weak scaling, no data-path
collective); the only collective is the final gather of the last step's depth images to rank 0 over
RCCL/xGMI, inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--image-size", type=int, default=512)
    ap.add_argument("--faces-per-pixel", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def build_batch(n_meshes, seed, device):
    import _util as U
    import pytorch3d_amd as p3d

    verts, faces = U.hetero_batch(n_meshes, seed=seed)
    nfaces = [int(f.shape[0]) for f in faces]
    meshes = p3d.PackedMeshes([v.to(device) for v in verts], [f.to(device) for f in faces])
    return meshes, verts, faces, nfaces


def cpu_baseline(verts, faces, H, W, K, blur):
    """Reference CPU path (oracle/_ref, the reference's own C++ kernels) on a bounded sample:
    the smallest mesh of the batch, full resolution, forward + backward."""
    from oracle import oracle as orc

    j = min(range(len(faces)), key=lambda i: faces[i].shape[0])
    v, f = verts[j], faces[j]
    fv = v[f].contiguous()
    F = fv.shape[0]
    first = torch.zeros(1, dtype=torch.int64)
    count = torch.tensor([F], dtype=torch.int64)
    nbr = torch.full((F,), -1, dtype=torch.int64)
    gen = torch.Generator().manual_seed(231)
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    ref = orc.ref_module()
    t0 = time.perf_counter()
    if ref is not None:
        kind = "reference"
        out = ref._rasterize_meshes_naive(fv, first, count, nbr, (H, W), blur, K, True, True, False)
        g = [torch.randn(o.shape, generator=gen) for o in out[1:]]
        ref.rasterize_meshes_backward(fv, out[0], g[0], g[1], g[2], True, True)
    else:
        kind = "port"
        out = orc.rasterize_meshes_naive(fv, first, count, nbr, (H, W), blur, K, True, True, False)
        g = [torch.randn(o.shape, generator=gen) for o in out[1:]]
        orc.rasterize_meshes_backward(fv, out[0], g[0], g[1], g[2], True, True)
    dt = time.perf_counter() - t0
    return {
        "value": H * W / dt / 1e6,
        "unit": "Mpix/s",
        "cores": cores,
        "kind": kind,
        "sample": f"1 mesh of the batch ({F} faces), {H}x{W}, K={K}, naive fwd (multi-thread) + bwd (single-thread, "
                  f"as the reference CPU path is), {dt:.1f} s",
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X); no CPU fallback exists for the product path")
    # P3D_BENCH_TEST_BACKEND=gloo maps every rank to cuda:0 and uses gloo: lets the multi-rank code path (barriers,
    # max-over-ranks timing, final gather) be exercised on a single-GPU box.  Never set by the driver.
    test_backend = os.environ.get("P3D_BENCH_TEST_BACKEND")
    if test_backend:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist_on = world > 1
    if dist_on:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if test_backend:
            dist.init_process_group(test_backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import pytorch3d_amd as p3d
    from pytorch3d_amd import _lib
    from pytorch3d_amd import sharding

    lib = _lib.load()
    H = W = args.image_size
    K = args.faces_per_pixel
    sigma = 1e-4
    import math

    blur = math.log(1.0 / 1e-4 - 1.0) * sigma  # SoftRas convention, tests/test_render_meshes.py:462
    B = args.batch
    meshes, verts_cpu, faces_cpu, nfaces = build_batch(B, seed=rank, device=device)
    total_faces = sum(nfaces)
    verts_packed = meshes.verts_packed().clone().requires_grad_(True)
    gen = torch.Generator().manual_seed(231 + rank)
    g_z = torch.randn((B, H, W, K), generator=gen).to(device)
    g_b = torch.randn((B, H, W, K, 3), generator=gen).to(device)
    g_d = torch.randn((B, H, W, K), generator=gen).to(device)

    def step():
        verts_packed.grad = None
        m = meshes.update_verts_packed(verts_packed)
        p2f, zbuf, bary, dists = p3d.rasterize_meshes(m, image_size=(H, W), blur_radius=blur, faces_per_pixel=K,
                                                      perspective_correct=True, clip_barycentric_coords=True)
        torch.autograd.backward([zbuf, bary, dists], [g_z, g_b, g_d])
        return zbuf

    def final_gather(z):
        # the one collective of the job: the final depth images of every rank are gathered on rank 0
        shard = z[..., 0].detach().contiguous()
        try:
            return sharding.gather_batch(shard, [B] * world, dst=0)
        except (RuntimeError, NotImplementedError):  # a backend without gather: every rank raises alike
            return sharding.gather_batch(shard, [B] * world)

    for _ in range(args.warmup):
        zbuf = step()
    if dist_on and args.warmup > 0:
        # part of the warmup: RCCL opens its point-to-point xGMI channels on the first gather (lazily, ~100 ms)
        del_me = final_gather(zbuf)
        del del_me
    torch.cuda.synchronize()
    lib.p3d_profile_reset()
    lib.p3d_profile_enable(1)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        zbuf = step()
    if dist_on:
        final = final_gather(zbuf)
        del final
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    t1 = time.perf_counter()
    lib.p3d_profile_enable(0)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=device)
    if dist_on:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    prof = _lib.profile_snapshot()
    hit_frac = float((zbuf >= 0).float().mean().item())

    if rank == 0:
        pixels = world * B * H * W * args.steps
        value = pixels / elapsed / 1e6
        # ---- roofline of the dominant kernel (algorithmic bytes) -------------------
        px = B * H * W
        alg = {
            "mesh_fine": px * K * 28 + total_faces * 44 + 16 * B,
            "mesh_naive": px * K * 28 + total_faces * 44 + 16 * B,
            "mesh_backward": px * K * 28 + 2 * total_faces * 36,
        }
        kernels = {k: {"launches": n, "avg_ms": ms / n} for k, (n, ms) in prof.items()}
        dom = max((k for k in kernels if k in alg), key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches"])
        achieved = alg[dom] / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom)
            except Exception:
                traffic = None
        roofline = {
            "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "algorithmic_bytes_per_launch": alg[dom],
            "avg_launch_ms": kernels[dom]["avg_ms"],
        }
        out = {
            "metric": "rasterized Mpix/s (fwd+bwd) at 512^2 faces_per_pixel=8",
            "value": value,
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[2]: batch of {B} heterogeneous meshes per GPU (1k-20k faces, tori/icospheres), "
                            "512x512, faces_per_pixel=8, SoftRas blur, perspective-correct + clipped bary, fwd+bwd",
                "global_batch": world * B, "image_size": [H, W], "faces_per_pixel": K, "blur_radius": blur,
                "total_faces_per_rank": total_faces, "pixel_slot_fill": hit_frac,
                "parallelism": f"batch-sharded x{world}, final gather to rank 0 only",
            },
            "roofline": roofline,
            "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in sorted(kernels.items())},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(verts_cpu, faces_cpu, H, W, K, blur)
            except Exception as e:  # the baseline must never sink the GPU measurement
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
