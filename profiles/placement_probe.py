#!/usr/bin/env python
"""Does the placement of the four output tensors move `mesh_fine`?  (profiles/r06/c31_33: the SAME library differs by up to 3 % between two
instances in one process.)  The forward of the bench batch over the C ABI with its outputs carved from ONE buffer at chosen relative
offsets; per pattern the library's own HIP-event time of `mesh_fine`, patterns interleaved over several rounds.

    python profiles/placement_probe.py [--iters 40] [--rounds 3]
"""
import argparse
import ctypes
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "profiles"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    import _util as U
    import exp_measure as E
    import pytorch3d_amd as p3d
    from pytorch3d_amd import _lib

    d = torch.device("cuda:0")
    B, H, K = 64, 512, 8
    blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
    verts, faces = U.hetero_batch(B, seed=0, torus_div=1.0)
    m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
    fv = m.verts_packed()[m.faces_packed()].contiguous()
    F = int(fv.shape[0])
    first, count = m.mesh_to_faces_packed_first_idx().contiguous(), m.num_faces_per_mesh().contiguous()
    nbr = torch.full((F,), -1, dtype=torch.int64, device=d)
    lib = _lib.load()
    bin_size, M = 32, int(max(10000, F / 5))
    ws = torch.empty((int(lib.p3d_rasterize_meshes_workspace_bytes(F, B, H, H, bin_size, M)),), dtype=torch.uint8, device=d)
    cover = torch.empty((B, H // 16, H // 16), dtype=torch.int32, device=d)
    stream = ctypes.c_void_p(torch.cuda.current_stream(d).cuda_stream)
    n = B * H * H * K
    sizes = [n * 8, n * 4, n * 12, n * 4]  # pix_to_face, zbuf, bary, dists
    slack = 64 << 20
    big = torch.empty((sum(sizes) + 8 * slack,), dtype=torch.uint8, device=d)
    base = (-big.data_ptr()) % (2 << 20)  # 2 MB-align the start

    def carve(gaps):
        ptrs, off = [], base
        for sz, g in zip(sizes, gaps):
            off += g
            ptrs.append(big.data_ptr() + off)
            off += sz
            off += (-off) % 256
        return ptrs

    KB, MB = 1 << 10, 1 << 20
    patterns = {
        "packed (as torch: 2 MB-aligned, back to back)": [0, 0, 0, 0],
        "+256 B each": [0, 256, 256, 256],
        "+4 KB each": [0, 4 * KB, 4 * KB, 4 * KB],
        "+64 KB each": [0, 64 * KB, 64 * KB, 64 * KB],
        "+1 MB + 4 KB each": [0, MB + 4 * KB, MB + 4 * KB, MB + 4 * KB],
        "+17 MB, +33 MB, +49 MB": [0, 17 * MB, 33 * MB, 49 * MB],
        "start +1 MB, then packed": [MB, 0, 0, 0],
    }

    def run(ptrs):
        rc = lib.p3d_rasterize_meshes_with_cover(fv.data_ptr(), first.data_ptr(), count.data_ptr(), nbr.data_ptr(), F, B, H, H, blur, K, bin_size,
                                                 M, 1, 1, 0, ptrs[0], ptrs[1], ptrs[2], ptrs[3], cover.data_ptr(), ws.data_ptr(), ws.numel(), stream)
        assert rc == 0, rc

    for _ in range(60):  # clocks
        run(carve(patterns["packed (as torch: 2 MB-aligned, back to back)"]))
    torch.cuda.synchronize()
    res = {k: [] for k in patterns}
    for r in range(args.rounds):
        order = list(patterns) if r % 2 == 0 else list(patterns)[::-1]
        for name in order:
            ptrs = carve(patterns[name])
            for _ in range(3):
                run(ptrs)
            torch.cuda.synchronize()
            lib.p3d_profile_reset()
            lib.p3d_profile_enable(1)
            for _ in range(args.iters):
                run(ptrs)
            torch.cuda.synchronize()
            lib.p3d_profile_enable(0)
            res[name].append(round(E.snapshot(lib)["mesh_fine"], 4))
    for name, v in res.items():
        print(f"{name:<52} mesh_fine ms per round: {v}   mean {sum(v) / len(v):.4f}")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
