#!/usr/bin/env python
"""Per-workgroup timeline of mesh_fine on the bench batch (needs a -DP3D_FWD_TIMELINE build):

    P3D_EXTRA_FLAGS="-DP3D_ABLATION -DP3D_FWD_TIMELINE" P3D_LIB_PATH=$PWD/pytorch3d_amd/libp3d_tl.so python -m pytorch3d_amd.build
    P3D_LIB_PATH=$PWD/pytorch3d_amd/libp3d_tl.so python profiles/fine_timeline.py

Every workgroup records s_memrealtime (100 MHz) at its start and end and its face count.  Printed: how many workgroups are
still running as a function of time (the tail), duration statistics by face-count class, and the sum of workgroup
durations per class (where the slot time goes)."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import _util as U
    import pytorch3d_amd as p3d
    from pytorch3d_amd import _C

    d = torch.device("cuda:0")
    verts, faces = U.hetero_batch(64, seed=0)
    m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
    fv = m.verts_packed()[m.faces_packed()].contiguous()
    nbr = torch.full((fv.shape[0],), -1, dtype=torch.int64, device=d)
    blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
    args = (fv, m.mesh_to_faces_packed_first_idx(), m.num_faces_per_mesh(), nbr, (512, 512), blur, 8, 32, 10000, True, True, False)
    for _ in range(3):
        _C.rasterize_meshes(*args)
    torch.cuda.synchronize()
    out = "/tmp/fine_timeline.bin"
    os.environ["P3D_FWD_TIMELINE_OUT"] = out
    _C.rasterize_meshes(*args)
    torch.cuda.synchronize()
    del os.environ["P3D_FWD_TIMELINE_OUT"]
    t = np.fromfile(out, dtype=np.uint64).reshape(-1, 3)
    t = t[t[:, 0] > 0]
    start = t[:, 0].astype(np.float64)
    end = t[:, 1].astype(np.float64)
    cnt = t[:, 2].astype(np.int64)
    t0 = start.min()
    start = (start - t0) / 100.0  # us
    end = (end - t0) / 100.0
    dur = end - start
    total = end.max()
    print(f"workgroups {len(t)}, kernel span {total:.1f} us, sum of workgroup durations {dur.sum() / 1e3:.1f} ms "
          f"(= {dur.sum() / total:.0f} workgroups in flight on average; 1024 slots)")
    for lo, hi in ((0, 0), (1, 64), (65, 256), (257, 512), (513, 1024), (1025, 10 ** 9)):
        s = (cnt >= lo) & (cnt <= hi)
        if s.any():
            print(f"  faces {lo:>5}..{hi if hi < 10**9 else 'inf':>5}: {int(s.sum()):6d} WGs, duration mean {dur[s].mean():7.1f} max {dur[s].max():7.1f} us, "
                  f"slot time {dur[s].sum() / 1e3:7.2f} ms ({100 * dur[s].sum() / dur.sum():4.1f} %)")
    for frac in (0.5, 0.8, 0.9, 0.95, 0.99, 1.0):
        print(f"  {100 * frac:5.1f} % of the workgroups have finished by {np.quantile(end, frac):7.1f} us")
    edges = np.linspace(0, total, 21)
    running = [(int(((start <= x) & (end > x)).sum())) for x in edges[:-1] + np.diff(edges) / 2]
    print("  workgroups running at 20 sample times:", running)
    order = np.argsort(-dur)[:8]
    print("  longest workgroups (start, duration us, faces):", [(round(float(start[i]), 1), round(float(dur[i]), 1), int(cnt[i])) for i in order])


if __name__ == "__main__":
    main()
