#!/usr/bin/env python
"""BASELINE configs[1] (the reference's cow, 256x256, K=8, blur 1e-4, coarse + fine forward): kernel milliseconds with
the split fine kernel (one workgroup per 8x8 sub-tile, the list dealt to its four waves) and without it.
Needs an ablation build:  P3D_LIB_PATH=$PWD/pytorch3d_amd/libp3d_abl.so python profiles/c2_split.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import _util as U
    from pytorch3d_amd import _C, _lib

    lib = _lib.load()
    d = torch.device("cuda:0")
    g = np.load(os.path.join(ROOT, "tests", "golden", "cow_ref.npz"))
    fv = torch.from_numpy(g["verts_ndc"])[torch.from_numpy(g["faces"]).long()].contiguous().to(d)
    F = fv.shape[0]
    cases = {"cow 256^2 K=8 blur 1e-4": (fv, torch.zeros(1, dtype=torch.int64, device=d), torch.tensor([F], device=d),
                                         torch.full((F,), -1, dtype=torch.int64, device=d), (256, 256), 1e-4, 8, 16, 10000, True, True, False)}
    v, f = U.ico_sphere(4)
    fs = U.to_ndc(v)[f].to(d).contiguous()
    cases["ico_sphere(4) 256^2 K=8 blur 1e-4"] = (fs, torch.zeros(1, dtype=torch.int64, device=d), torch.tensor([fs.shape[0]], device=d),
                                                  torch.full((fs.shape[0],), -1, dtype=torch.int64, device=d), (256, 256), 1e-4, 8, 16, 10000, True, True, False)
    for name, args in cases.items():
        ref = None
        for mode, dbg in (("split", "0"), ("one workgroup per 16x16 tile", "1024")):
            os.environ["P3D_DEBUG_FWD"] = dbg
            for _ in range(5):
                out = _C.rasterize_meshes(*args)
            torch.cuda.synchronize()
            lib.p3d_profile_reset()
            lib.p3d_profile_enable(1)
            t0 = time.perf_counter()
            for _ in range(50):
                out = _C.rasterize_meshes(*args)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 50 * 1e3
            lib.p3d_profile_enable(0)
            prof = {k: round(ms / n, 4) for k, (n, ms) in _lib.profile_snapshot().items()}
            if ref is None:
                ref = out
            same = all(torch.equal(a, b) for a, b in zip(out, ref))
            print(f"{name} | {mode}: wall {wall:.4f} ms, mesh_fine {prof.get('mesh_fine')} ms, kernels {sum(prof.values()):.4f} ms, identical: {same}", flush=True)
    os.environ.pop("P3D_DEBUG_FWD", None)


if __name__ == "__main__":
    main()
