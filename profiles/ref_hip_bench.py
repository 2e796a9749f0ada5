#!/usr/bin/env python
"""Same-GPU timing of the REFERENCE's own device kernels (hipified as a checker, oracle/build_ref_hip.py) next to ours.

    python profiles/ref_hip_bench.py  ->  one JSON line (copy into profiles/rNN_vs_reference_hip.json)

The reference's kernels are what a PyTorch3D user gets on this GPU from a ROCm build of pytorch3d (torch hipify + hipcc,
default flags).  Same inputs, same operator boundary (`_C.rasterize_meshes` / `rasterize_meshes_backward`, etc.), timed
with torch.cuda events on the current stream (both libraries launch there), median of `iters`.
config 3 = the bench batch (64 meshes, 512^2, K=8, SoftRas blur), bin_size / max_faces_per_bin as the reference's
Python wrapper picks them (renderer/mesh/rasterize_meshes.py:201-222: 32 and max(10000, F/5)).
"""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def med_ms(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    import _util as U
    import pytorch3d_amd as p3d
    from oracle import oracle as orc
    from pytorch3d_amd import _C

    ref = orc.ref_hip_module(nofma=False)
    if ref is None:
        raise SystemExit("oracle/_ref/p3d_ref_hip.so not built")
    d = torch.device("cuda:0")
    out = {}
    # ---- config 3 ------------------------------------------------------------------------------------------
    B, H, K = 64, 512, 8
    blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
    verts, faces = U.hetero_batch(B, seed=0)
    m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
    fv = m.verts_packed()[m.faces_packed()].contiguous()
    F = fv.shape[0]
    first, cnt = m.mesh_to_faces_packed_first_idx(), m.num_faces_per_mesh()
    nbr = torch.full((F,), -1, dtype=torch.int64, device=d)
    M = int(max(10000, F / 5))
    args = (fv, first, cnt, nbr, (H, H), blur, K, 32, M, True, True, False)
    ours = _C.rasterize_meshes(*args)
    theirs = ref.rasterize_meshes(*args)
    same = float((ours[0] == theirs[0]).float().mean())
    gen = torch.Generator().manual_seed(231)
    gz = torch.randn((B, H, H, K), generator=gen).to(d)
    gb = torch.randn((B, H, H, K, 3), generator=gen).to(d)
    gd = torch.randn((B, H, H, K), generator=gen).to(d)
    t_of = med_ms(lambda: _C.rasterize_meshes(*args))
    t_rf = med_ms(lambda: ref.rasterize_meshes(*args), iters=7, warm=1)
    t_ob = med_ms(lambda: _C.rasterize_meshes_backward(fv, ours[0], gz, gb, gd, True, True))
    t_rb = med_ms(lambda: ref.rasterize_meshes_backward(fv, ours[0], gz, gb, gd, True, True), iters=7, warm=1)
    px = B * H * H / 1e6
    out["config3_batch64_512_k8"] = {
        "faces": F, "max_faces_per_bin": M, "pix_to_face_agreement": same,
        "ours_ms": {"forward": t_of, "backward": t_ob}, "reference_hip_ms": {"forward": t_rf, "backward": t_rb},
        "ours_Mpix_s_fwd_bwd": px / ((t_of + t_ob) * 1e-3), "reference_Mpix_s_fwd_bwd": px / ((t_rf + t_rb) * 1e-3),
        "speedup_forward": t_rf / t_of, "speedup_backward": t_rb / t_ob, "speedup_fwd_bwd": (t_rf + t_rb) / (t_of + t_ob)}
    del theirs, gz, gb, gd
    torch.cuda.empty_cache()
    # ---- config 2: the cow ---------------------------------------------------------------------------------
    g = np.load(os.path.join(ROOT, "tests", "golden", "cow_ref.npz"))
    fvc = torch.from_numpy(g["verts_ndc"])[torch.from_numpy(g["faces"]).long()].contiguous().to(d)
    Fc = fvc.shape[0]
    a2 = (fvc, torch.zeros(1, dtype=torch.int64, device=d), torch.tensor([Fc], device=d), torch.full((Fc,), -1, dtype=torch.int64, device=d),
          (256, 256), 1e-4, 8, 16, 10000, True, True, False)
    out["config2_cow_256_k8_fwd"] = {"ours_ms": med_ms(lambda: _C.rasterize_meshes(*a2)),
                                     "reference_hip_ms": med_ms(lambda: ref.rasterize_meshes(*a2))}
    # ---- config 4: 1M points + alpha compositor ------------------------------------------------------------
    gen = torch.Generator().manual_seed(0)
    P, Hp, Kp, r, C = 1_000_000, 512, 10, 0.01, 3
    pts = torch.cat([torch.rand(P, 2, generator=gen) * 2 - 1, torch.rand(P, 1, generator=gen) * 2 + 0.5], 1).to(d)
    feats = torch.rand(C, P, generator=gen).to(d)
    pf = torch.zeros(1, dtype=torch.int64, device=d)
    pc = torch.full((1,), P, dtype=torch.int64, device=d)
    rad = torch.full((P,), r, device=d)
    a4 = (pts, pf, pc, (Hp, Hp), rad, Kp, 32, 200000)
    idx, zb, ds = _C.rasterize_points(*a4)
    gz = torch.randn(zb.shape, generator=gen).to(d)
    gd = torch.randn(zb.shape, generator=gen).to(d)
    al = (1 - ds / (r * r)).clamp(0, 1).permute(0, 3, 1, 2).contiguous()
    pi = idx.long().permute(0, 3, 1, 2).contiguous()
    gi = torch.randn((1, C, Hp, Hp), generator=gen).to(d)
    res = {}
    for tag, mod in (("ours_ms", _C), ("reference_hip_ms", ref)):
        res[tag] = {
            "rasterize_points": med_ms(lambda: mod.rasterize_points(*a4), iters=7 if mod is ref else 20, warm=1),
            "rasterize_points_backward": med_ms(lambda: mod.rasterize_points_backward(pts, idx, gz, gd)),
            "alpha_composite": med_ms(lambda: mod.accum_alphacomposite(feats, al, pi)),
            "alpha_composite_backward": med_ms(lambda: mod.accum_alphacomposite_backward(gi, feats, al, pi)),
        }
        res[tag]["total"] = sum(res[tag].values())
    res["speedup_total"] = res["reference_hip_ms"]["total"] / res["ours_ms"]["total"]
    out["config4_points_1m_512_k10_fwd_bwd"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
