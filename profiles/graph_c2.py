#!/usr/bin/env python
"""BASELINE config 2 (one 5k-face mesh, 256x256, K=8, forward) replayed from a HIP graph: the C ABI launches are
asynchronous on the caller's stream, allocate nothing and never synchronise, so `torch.cuda.graph` captures them as
they are.  Prints eager vs graph-replay wall time per call.  Run on the GPU box:  python profiles/graph_c2.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import _util as U
    from pytorch3d_amd import _C

    d = torch.device("cuda:0")
    v, f = U.ico_sphere(4)
    fv = U.to_ndc(v)[f].to(d).contiguous()
    F = fv.shape[0]
    first = torch.zeros(1, dtype=torch.int64, device=d)
    cnt = torch.tensor([F], dtype=torch.int64, device=d)
    nbr = torch.full((F,), -1, dtype=torch.int64, device=d)
    args = (fv, first, cnt, nbr, (256, 256), 1e-4, 8, 16, 10000, True, True, False)

    def timeit(fn, iters=200):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    eager = timeit(lambda: _C.rasterize_meshes(*args))
    ref = _C.rasterize_meshes(*args)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            _C.rasterize_meshes(*args)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = _C.rasterize_meshes(*args)
    replay = timeit(g.replay)
    g.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(out, ref))
    print(f"config 2 forward: eager {eager:.4f} ms/call, graph replay {replay:.4f} ms/call, identical outputs: {same}", flush=True)


if __name__ == "__main__":
    main()
