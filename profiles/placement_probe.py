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
    ap.add_argument("--only-allocations", action="store_true", help="skip the layout patterns, the fills and the one-buffer swaps (for counter runs: "
                    "60 warm-up launches, then rounds x 6 allocations x (3 + iters) launches in that order)")
    ap.add_argument("--reserve-gb", type=float, default=0.0, help="hold a dummy allocation of this size, made before anything else")
    ap.add_argument("--image-size", type=int, default=512, help="square image side (512: a row of pix_to_face is 32 KiB, a power of two; 496: 31 KiB)")
    args = ap.parse_args()
    import _util as U
    import exp_measure as E
    import pytorch3d_amd as p3d
    from pytorch3d_amd import _lib

    d = torch.device("cuda:0")
    reserve = torch.empty((int(args.reserve_gb * (1 << 30)),), dtype=torch.uint8, device=d) if args.reserve_gb > 0 else None  # noqa: F841
    # the very first allocations of the process: one set of outputs (3.76 GB + 4 KB) and, behind it, a workspace-sized block
    n0 = 64 * args.image_size * args.image_size * 8
    first_out = torch.empty((n0 * 28 + 4096,), dtype=torch.uint8, device=d)
    B, H, K = 64, args.image_size, 8
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
    cover = torch.empty((B, (H + 15) // 16, (H + 15) // 16), dtype=torch.int32, device=d)
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
    for r in range(0 if args.only_allocations else args.rounds):
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
        if v:
            print(f"{name:<52} mesh_fine ms per round: {v}   mean {sum(v) / len(v):.4f}")
    print(json.dumps(res))

    # Second question: do two independent ALLOCATIONS differ (physical pages), with the same code and the same relative layout?
    # Six allocations of outputs + workspace + cover, made one after the other and all kept alive; measured in turn, three rounds.
    sets = [(first_out, torch.empty_like(ws), torch.empty_like(cover))]
    for i in range(5):
        o = torch.empty((sum(sizes) + 4096,), dtype=torch.uint8, device=d)
        w = torch.empty_like(ws)
        c = torch.empty_like(cover)
        sets.append((o, w, c))
        _pad = torch.empty((37 << 20) * (i + 1), dtype=torch.uint8, device=d)  # (shifts the next allocation; freed at once)
        del _pad
    res2 = {i: [] for i in range(len(sets))}
    for r in range(3):
        for i, (o, w, c) in enumerate(sets):
            off, ptrs = (-o.data_ptr()) % 256, []
            for sz in sizes:
                ptrs.append(o.data_ptr() + off)
                off += sz
            def go():
                rc = lib.p3d_rasterize_meshes_with_cover(fv.data_ptr(), first.data_ptr(), count.data_ptr(), nbr.data_ptr(), F, B, H, H, blur, K,
                                                         bin_size, M, 1, 1, 0, ptrs[0], ptrs[1], ptrs[2], ptrs[3], c.data_ptr(), w.data_ptr(),
                                                         w.numel(), stream)
                assert rc == 0, rc
            for _ in range(3):
                go()
            torch.cuda.synchronize()
            lib.p3d_profile_reset()
            lib.p3d_profile_enable(1)
            for _ in range(args.iters):
                go()
            torch.cuda.synchronize()
            lib.p3d_profile_enable(0)
            res2[i].append(round(E.snapshot(lib)["mesh_fine"], 4))
    # Third: which of the three buffers carries the effect?  One buffer from allocation i, the other two from allocation 0.
    def measure(o, w, c):
        off, ptrs = (-o.data_ptr()) % 256, []
        for sz in sizes:
            ptrs.append(o.data_ptr() + off)
            off += sz
        def go():
            rc = lib.p3d_rasterize_meshes_with_cover(fv.data_ptr(), first.data_ptr(), count.data_ptr(), nbr.data_ptr(), F, B, H, H, blur, K,
                                                     bin_size, M, 1, 1, 0, ptrs[0], ptrs[1], ptrs[2], ptrs[3], c.data_ptr(), w.data_ptr(),
                                                     w.numel(), stream)
            assert rc == 0, rc
        for _ in range(3):
            go()
        torch.cuda.synchronize()
        lib.p3d_profile_reset()
        lib.p3d_profile_enable(1)
        for _ in range(args.iters):
            go()
        torch.cuda.synchronize()
        lib.p3d_profile_enable(0)
        return round(E.snapshot(lib)["mesh_fine"], 4)

    if args.only_allocations:
        for i, v in res2.items():
            print(f"allocation {i}: mesh_fine ms per round {v}")
        return
    # Is a slow allocation slow for a plain streaming fill too?  (then: a property of its pages; else: of the kernel's write ORDER on them)
    fills = []
    for o, _, _ in sets:
        for _ in range(2):
            o.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            o.zero_()
        e1.record()
        torch.cuda.synchronize()
        fills.append(round(e0.elapsed_time(e1) / 10, 4))
    print(f"streaming fill of the outputs of allocation i = 0..5 ({sum(sizes) / 1e9:.2f} GB), ms: {fills}")
    res3 = {"outputs": [], "workspace": [], "cover": []}
    for i in range(len(sets)):
        res3["outputs"].append(measure(sets[i][0], sets[0][1], sets[0][2]))
        res3["workspace"].append(measure(sets[0][0], sets[i][1], sets[0][2]))
        res3["cover"].append(measure(sets[0][0], sets[0][1], sets[i][2]))
    for k, v in res3.items():
        print(f"only the {k} from allocation i = 0..5 (rest from allocation 0): {v}")
    print(json.dumps({"one_buffer_swapped": res3}))
    for i, v in res2.items():
        print(f"allocation {i}{' (outputs allocated FIRST in the process)' if i == 0 else ''} (outputs at {sets[i][0].data_ptr():#x}, workspace at {sets[i][1].data_ptr():#x}): mesh_fine ms per round {v}")
    print(json.dumps({"allocations": res2}))


if __name__ == "__main__":
    main()
