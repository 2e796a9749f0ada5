#!/usr/bin/env python
"""BASELINE configs[1] (the reference's cow, 256x256, K=8, blur 1e-4, coarse + fine forward) on the product library:
wall per call without instrumentation, then the per-kernel HIP-event times of the same calls."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from pytorch3d_amd import _C, _lib

    lib = _lib.load()
    d = torch.device("cuda:0")
    g = np.load(os.path.join(ROOT, "tests", "golden", "cow_ref.npz"))
    fv = torch.from_numpy(g["verts_ndc"])[torch.from_numpy(g["faces"]).long()].contiguous().to(d)
    F = fv.shape[0]
    args = (fv, torch.zeros(1, dtype=torch.int64, device=d), torch.tensor([F], device=d),
            torch.full((F,), -1, dtype=torch.int64, device=d), (256, 256), 1e-4, 8, 16, 10000, True, True, False)
    for _ in range(20):
        _C.rasterize_meshes(*args)
    torch.cuda.synchronize()
    walls = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(200):
            _C.rasterize_meshes(*args)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) / 200 * 1e3)
    lib.p3d_profile_reset()
    lib.p3d_profile_enable(1)
    for _ in range(50):
        _C.rasterize_meshes(*args)
    torch.cuda.synchronize()
    lib.p3d_profile_enable(0)
    prof = {k: round(ms / n, 4) for k, (n, ms) in _lib.profile_snapshot().items()}
    print(f"cow 256^2 K=8 blur 1e-4: wall {min(walls):.4f} ms (median {sorted(walls)[2]:.4f}), kernels {sum(prof.values()):.4f} ms: {prof}")


if __name__ == "__main__":
    main()
