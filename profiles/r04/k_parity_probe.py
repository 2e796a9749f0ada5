#!/usr/bin/env python
"""Mismatch counts of rasterize_meshes vs the C oracle per K on the overflowing soup of tests/test_gpu_meshes.py::test_all_queue_capacities."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _util as U
from oracle import oracle as orc
from pytorch3d_amd import _C

d = torch.device("cuda:0")
for K in [int(x) for x in sys.argv[1:]]:
    gen = torch.Generator().manual_seed(K)
    U.triangle_soup(300, gen, size=1.0)
    big = U.triangle_soup(260, gen, size=4.0)
    first, count = U.split_counts(260, 1)
    nbr = torch.full((260,), -1, dtype=torch.int64)
    for persp, clip in ((True, True), (False, False)):
        ref = orc.rasterize_meshes_naive(big, first, count, nbr, (20, 37), 0.02, K, persp, clip, False)
        for bs in (0, 8):
            o = _C.rasterize_meshes(big.to(d), first.to(d), count.to(d), nbr.to(d), (20, 37), 0.02, K, bs, 300 if bs else 0, persp, clip, False)
            o = [t.cpu() for t in o]
            bad = (o[0] != ref[0])
            fl = [int((a != b).sum()) for a, b in zip(o[1:], ref[1:])]
            first_bad = bad.nonzero()[0].tolist() if bad.any() else None
            print(f"K={K} persp={persp} clip={clip} bin={bs}: p2f mismatches {int(bad.sum())} floats {fl} first {first_bad}"
                  + (f" ours {o[0][tuple(first_bad[:3])].tolist()} ref {ref[0][tuple(first_bad[:3])].tolist()}" if first_bad else ""), flush=True)
