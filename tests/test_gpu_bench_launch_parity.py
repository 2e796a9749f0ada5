"""The EXACT launch `bench.py` times, against the reference's own device kernels on the same MI355X.

`bench.py` (BASELINE configs[2]) rasterizes `hetero_batch(64, seed=0, torus_div=1.0)` (SURVEY.md 8(d) config 3 as written;
the lighter torus_div=1.5 batch of rounds 1-3 is the second parameter) at 512x512, K = 8, SoftRas blur, perspective-correct
+ clipped barycentrics: one `mesh_fine` launch of 65,536 workgroups over 64 images (41k background tiles, the XCD tile
map over the full grid, output offsets past 2^31 bytes) and one `mesh_backward` launch.  The smaller parity tests
(tests/test_gpu_baseline_sizes.py, test_gpu_vs_reference_device_kernels.py) rasterize four of the 64 meshes; this one
runs the whole batch through `_C.rasterize_meshes` and through oracle/_ref/p3d_ref_hip_nofma.so (the reference's .cu files
hipified and compiled with -ffp-contract=off, oracle/build_ref_hip.py; ~2 s per forward) and asserts, on the GPU:

  * zbuf bit-equal everywhere; bary / dists bit-equal wherever pix_to_face agrees;
  * pix_to_face differences only in slots whose depth ties exactly with a neighbouring slot of the pixel (or the last
    slot, whose tie partner was evicted): the reference's CUDA queue orders by z alone, ours by (z, index) like the
    reference's CPU / Python implementations (SURVEY appendix A "Top-K"); the count is printed;
  * the backward of the same fragments: every entry within 5e-3 (the reference's own gradient tolerance,
    tests/test_rasterize_meshes.py:317-319) x the sum of the absolute per-sample terms of the float64 restatement of the
    reference's formulas (oracle/backward_f64.py), and within twice that of the reference's device backward wherever
    the latter passes the same gate.  (A tolerance scaled by the largest gradient of the batch -- 1e24, faces seen edge-on
    -- was vacuous; the per-face comparison that replaced it in round 3 found gradients of 1e15 where the reference has
    0, a reciprocal estimate breaking the reference's exact cancellation `1 / s - w / s^2`: fixed in csrc/p3d_geom.h
    bary_clip_bwd, and the cases where the reference's own value is that residue times 1e16 are now counted.)

The same inputs through the L2 mirror (`pytorch3d_amd.rasterize_meshes`, what bench.py calls) must give the same bits
as the `_C` operator (the mirror bins at its own heuristic bin_size; results do not depend on the binning).
"""
import math

import pytest
import torch

import _util as U
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

SOFTRAS_BLUR = math.log(1.0 / 1e-4 - 1.0) * 1e-4
B, H, K = 64, 512, 8


def _bits(t):
    return t.view(torch.int32)


# (the lighter torus_div = 1.5 batch of rounds 1-3 is no longer a parameter: 22 s of the suite for a workload nothing quotes)
@pytest.mark.parametrize("torus_div", [U.CONFIG3_TORUS_DIV], ids=["config3_literal"])
def test_full_bench_batch_forward_and_backward_vs_reference_device_kernels(torus_div):
    mod = orc.ref_hip_module(nofma=True)
    if mod is None:
        pytest.skip("oracle/_ref/p3d_ref_hip_nofma.so not built (oracle/build_ref_hip.py, build container only)")
    import pytorch3d_amd as p3d
    from pytorch3d_amd import _C

    d = torch.device("cuda:0")
    # bench.py: build_batch(B, seed=rank), rank 0 -- torus_div 1.0 is the headline workload (SURVEY.md 8(d) config 3 as written:
    # tori unscaled, crossing the frame edge, ~58 % of the pixels covered), 1.5 the lighter batch rounds 1-3 quoted
    verts, faces = U.hetero_batch(B, seed=0, torus_div=torus_div)
    m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
    fv = m.verts_packed()[m.faces_packed()].contiguous()
    F = int(fv.shape[0])
    first, count = m.mesh_to_faces_packed_first_idx(), m.num_faces_per_mesh()
    nbr = torch.full((F,), -1, dtype=torch.int64, device=d)
    M = int(max(10000, F / 5))  # renderer/mesh/rasterize_meshes.py:216-222
    args = (fv, first, count, nbr, (H, H), SOFTRAS_BLUR, K, 32, M, True, True, False)
    ours = _C.rasterize_meshes(*args)
    theirs = mod.rasterize_meshes(*args)
    torch.cuda.synchronize()

    same = ours[0] == theirs[0]
    n_idx = int((~same).sum())
    total = same.numel()
    assert torch.equal(_bits(ours[1]), _bits(theirs[1])), "zbuf is not bit-equal on the bench batch"
    # bary / dists: bit-equal wherever the index agrees
    bad_d = int(((_bits(ours[3]) != _bits(theirs[3])) & same).sum())
    bad_b = int(((_bits(ours[2]) != _bits(theirs[2])).any(-1) & same).sum())
    z = ours[1]
    tie = torch.zeros_like(same)
    tie[..., 1:] |= z[..., 1:] == z[..., :-1]
    tie[..., :-1] |= z[..., :-1] == z[..., 1:]
    tie[..., K - 1] = True
    unexplained = int((~same & ~tie).sum())
    covered = float((ours[0][..., 0] >= 0).float().mean())
    print(f"[bench launch torus_div={torus_div}: {B} meshes, {F} faces, {H}^2, K={K}] pix_to_face differences {n_idx} / {total} "
          f"({n_idx / total:.2e}), not at an exact depth tie: {unexplained}; bary / dists words differing where the index "
          f"agrees: {bad_b} / {bad_d}; covered pixels {covered:.3f}")
    assert bad_d == 0 and bad_b == 0
    assert unexplained == 0
    assert n_idx <= 1e-3 * total  # observed: 1.7e-4 (torus_div 1.5); all of them at exact depth ties (asserted above)
    # (With `_C.CUDA_TIE_ORDER` our survivors at exact ties are those of the reference's procedure on faces in ascending index --
    # equal to its NAIVE device kernel bit for bit, tests/test_gpu_vs_reference_device_kernels.py; its binned path, compared here,
    # orders a bin's faces across 512-face chunks by atomicAdd arrival, rasterize_coarse.cu:185, and is not a fixed target at ties.)
    del theirs, tie, same
    torch.cuda.empty_cache()

    # the L2 mirror (what bench.py calls) == the _C operator, bit for bit
    out = p3d.rasterize_meshes(m, image_size=(H, H), blur_radius=SOFTRAS_BLUR, faces_per_pixel=K, perspective_correct=True,
                               clip_barycentric_coords=True)
    assert torch.equal(out[0], ours[0])
    for a, b in zip(out[1:], ours[1:]):
        assert torch.equal(_bits(a), _bits(b))
    del out
    torch.cuda.empty_cache()

    # backward on the same fragments
    gen = torch.Generator().manual_seed(231)  # bench.py's upstream gradients (rank 0)
    gz = torch.randn((B, H, H, K), generator=gen).to(d)
    gb = torch.randn((B, H, H, K, 3), generator=gen).to(d)
    gd = torch.randn((B, H, H, K), generator=gen).to(d)
    a = _C.rasterize_meshes_backward(fv, ours[0], gz, gb, gd, True, True)
    b = mod.rasterize_meshes_backward(fv, ours[0], gz, gb, gd, True, True)
    torch.cuda.synchronize()
    # gate: the float64 restatement of the reference's backward (oracle/backward_f64.py), error measured against the sum
    # of the absolute per-sample terms; the reference's device result is judged by the same gate and printed
    U.assert_face_grads_vs_truth("bench launch backward", a, fv, ours[0], gz, gb, gd, True, True, reference=b)
