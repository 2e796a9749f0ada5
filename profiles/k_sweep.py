#!/usr/bin/env python
"""rasterize_meshes forward + backward kernel times vs faces_per_pixel on 8 meshes of the bench generator (512x512).
Run on the GPU box:  python profiles/k_sweep.py 1 4 8 10 16 32 50 100"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import _util as U
    import pytorch3d_amd as p3d
    from pytorch3d_amd import _C, _lib

    d = torch.device("cuda:0")
    B = int(os.environ.get("ABL_BATCH", "8"))
    verts, faces = U.hetero_batch(B, seed=0)
    m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
    fv = m.verts_packed()[m.faces_packed()].contiguous()
    first, cnt = m.mesh_to_faces_packed_first_idx(), m.num_faces_per_mesh()
    nbr = torch.full((fv.shape[0],), -1, dtype=torch.int64, device=d)
    blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
    lib = _lib.load()
    gen = torch.Generator().manual_seed(1)
    for K in [int(x) for x in sys.argv[1:]] or [8]:
        out = _C.rasterize_meshes(fv, first, cnt, nbr, (512, 512), blur, K, 32, 10000, True, True, False)
        gz = torch.randn(B, 512, 512, K, generator=gen).to(d)
        gb = torch.randn(B, 512, 512, K, 3, generator=gen).to(d)
        gd = torch.randn(B, 512, 512, K, generator=gen).to(d)

        def step():
            o = _C.rasterize_meshes(fv, first, cnt, nbr, (512, 512), blur, K, 32, 10000, True, True, False)
            _C.rasterize_meshes_backward(fv, o[0], gz, gb, gd, True, True)

        step()
        torch.cuda.synchronize()
        lib.p3d_profile_reset()
        lib.p3d_profile_enable(1)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        lib.p3d_profile_enable(0)
        prof = _lib.profile_snapshot()
        fill = float((out[0] >= 0).float().mean())
        print(f"K={K}: mesh_fine {prof['mesh_fine'][1] / prof['mesh_fine'][0]:.3f} ms, mesh_backward "
              f"{prof['mesh_backward'][1] / prof['mesh_backward'][0]:.3f} ms, slot fill {fill:.3f}", flush=True)
        del out, gz, gb, gd


if __name__ == "__main__":
    main()
