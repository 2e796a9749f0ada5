#!/usr/bin/env python
"""What `_C.CUDA_TIE_ORDER` costs: the forward of the bench batch (BASELINE configs[2], config 3 as written) with and without it.

    python profiles/tie_order_timing.py [--batch 64] [K ...]       ->  one line per K

With the switch on, the fine kernel's TIES instantiation marks the pixels in which an entry may have been dropped at the depth of
the K-th survivor (csrc/raster_mesh.hip: eval_candidates) and `mesh_cuda_order_kernel` replays the reference's CUDA procedure
(rasterize_meshes.cu:216-237) for those pixels only.  Printed: kernel milliseconds of both forms from the library's HIP-event
profiler, the entries in which the two results differ, and the pixels the replay rewrote with other survivors."""
import argparse
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("K", nargs="*", type=int, default=[8])
    ap.add_argument("--count-marks", action="store_true", help="run with P3D_TIE_SKIP_REPLAY=1: the marks stay in the output and are counted")
    args = ap.parse_args()
    if args.count_marks:
        os.environ["P3D_TIE_SKIP_REPLAY"] = "1"
    import _util as U
    import pytorch3d_amd as p3d
    from pytorch3d_amd import _C, _lib

    d = torch.device("cuda:0")
    B = args.batch
    verts, faces = U.hetero_batch(B, seed=0, torus_div=U.CONFIG3_TORUS_DIV)
    m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
    fv = m.verts_packed()[m.faces_packed()].contiguous()
    first, cnt = m.mesh_to_faces_packed_first_idx(), m.num_faces_per_mesh()
    nbr = torch.full((fv.shape[0],), -1, dtype=torch.int64, device=d)
    blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
    lib = _lib.load()
    M = int(max(10000, fv.shape[0] / 5))
    for K in args.K:
        res = {}
        for mode in (False, True):
            _C.CUDA_TIE_ORDER = mode
            for _ in range(3):
                out = _C.rasterize_meshes(fv, first, cnt, nbr, (512, 512), blur, K, 32, M, True, True, False)
            torch.cuda.synchronize()
            lib.p3d_profile_reset()
            lib.p3d_profile_enable(1)
            for _ in range(10):
                out = _C.rasterize_meshes(fv, first, cnt, nbr, (512, 512), blur, K, 32, M, True, True, False)
            torch.cuda.synchronize()
            lib.p3d_profile_enable(0)
            prof = {k: ms / n for k, (n, ms) in _lib.profile_snapshot().items()}
            res[mode] = (prof, [o.clone() for o in out])
        _C.CUDA_TIE_ORDER = False
        plain, tied = res[False], res[True]
        if args.count_marks:
            marks = tied[1][0][..., 0] == -2
            waves = marks.view(B, 64, 8, 64, 8).any(-1).any(2)
            print(f"K={K} batch {B}: {int(marks.sum())} marked pixels of {marks.numel()} in {int(waves.sum())} of {waves.numel()} sub-tiles; "
                  f"TIES mesh_fine {tied[0]['mesh_fine']:.3f} ms (plain {plain[0]['mesh_fine']:.3f})", flush=True)
            continue
        diff = plain[1][0] != tied[1][0]
        assert int((tied[1][0] == -2).sum()) == 0, "a mark survived the replay"
        assert torch.equal(plain[1][1].view(torch.int32), tied[1][1].view(torch.int32)), "zbuf must not depend on the tie order"
        fine0 = plain[0]["mesh_fine"]
        fine1, rep = tied[0]["mesh_fine"], tied[0].get("mesh_cuda_order", 0.0)
        print(f"K={K} batch {B}: plain mesh_fine {fine0:.3f} ms | CUDA tie order: mesh_fine {fine1:.3f} + replay {rep:.3f} = "
              f"{fine1 + rep:.3f} ms ({(fine1 + rep) / fine0:.2f} x) | entries that differ {int(diff.sum())} of {diff.numel()}, "
              f"pixels {int(diff.any(-1).sum())}", flush=True)


if __name__ == "__main__":
    main()
