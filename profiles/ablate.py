#!/usr/bin/env python
"""Ablation timing of the two big kernels on the bench workload (run on the GPU box).

    python profiles/ablate.py bwd 0 1 2 3 4 7       # P3D_DEBUG_BWD values
    python profiles/ablate.py fwd 0 1 2 ...         # P3D_DEBUG_FWD values
Prints one line per variant: average kernel ms (HIP events inside the library).

The switches exist only in an ablation build (the product library never reads the environment):
    P3D_EXTRA_FLAGS="-DP3D_ABLATION -DP3D_FWD_STATS" P3D_LIB_PATH=$PWD/pytorch3d_amd/libp3d_amd_abl.so python -m pytorch3d_amd.build
    P3D_LIB_PATH=$PWD/pytorch3d_amd/libp3d_amd_abl.so python profiles/ablate.py fwd 0 1 2 64
"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    which = sys.argv[1]
    variants = sys.argv[2:] or ["0"]
    import _util as U
    import pytorch3d_amd as p3d
    from pytorch3d_amd import _C, _lib

    B = int(os.environ.get("ABL_BATCH", "64"))
    H = W = int(os.environ.get("ABL_SIZE", "512"))
    K = 8
    d = torch.device("cuda:0")
    verts, faces = U.hetero_batch(B, seed=0)
    m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
    fv = m.verts_packed()[m.faces_packed()].contiguous()
    first, cnt = m.mesh_to_faces_packed_first_idx(), m.num_faces_per_mesh()
    nbr = torch.full((fv.shape[0],), -1, dtype=torch.int64, device=d)
    blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
    lib = _lib.load()

    def fwd():
        return _C.rasterize_meshes(fv, first, cnt, nbr, (H, W), blur, K, 32, 10000, True, True, False)

    out = fwd()
    gen = torch.Generator().manual_seed(231)
    gz = torch.randn((B, H, W, K), generator=gen).to(d)
    gb = torch.randn((B, H, W, K, 3), generator=gen).to(d)
    gd = torch.randn((B, H, W, K), generator=gen).to(d)

    def bwd():
        return _C.rasterize_meshes_backward(fv, out[0], gz, gb, gd, True, True)

    fn, env, kern = (bwd, "P3D_DEBUG_BWD", "mesh_backward") if which == "bwd" else (fwd, "P3D_DEBUG_FWD", "mesh_fine")
    for v in variants:
        os.environ[env] = v
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        lib.p3d_profile_reset()
        lib.p3d_profile_enable(1)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        lib.p3d_profile_enable(0)
        prof = _lib.profile_snapshot()
        n, ms = prof[kern]
        print(f"{env}={v}: {kern} {ms / n:.3f} ms", flush=True)
    os.environ.pop(env, None)


if __name__ == "__main__":
    main()
