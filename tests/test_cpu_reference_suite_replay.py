"""CPU suite, part 2: the oracle replays every hot-path operator call that the REFERENCE'S OWN
TEST-SUITE makes on CPU (tests/test_rasterize_meshes.py, test_rasterize_points.py, test_compositing.py
of the reference, recorded while running on the reference's CPU kernels -- see
tests/golden/record_reference_suite.py).  The recorded outputs are the ones those tests compare with
their hand-written golden tensors (_simple_triangle_raster, blurry raster, perspective-correct 11x11,
back-face culling, 5x5 / 16x16 point goldens, the coarse bin goldens, the 4x4 compositing goldens), so
reproducing them pins the oracle to the reference's golden vectors.
"""
import pytest
import torch

import _util as U
from oracle import oracle as orc

CALLS = U.ref_suite_calls()
IDS = [f"{i}-{op}-{test.split('.')[-1]}" for i, (op, test, _, _) in enumerate(CALLS)]


def _close(a, b, atol, rtol=0.0):
    assert a.shape == b.shape
    assert torch.allclose(a, b, atol=atol, rtol=rtol), f"max diff {(a - b).abs().max().item()}"


@pytest.mark.parametrize("op,test,args,outs", CALLS, ids=IDS)
def test_oracle_replays_reference_suite_call(op, test, args, outs):
    if op == "rasterize_meshes":
        fv, first, count, nbr, size, blur, K, bin_size, M, persp, clip, cull = args
        assert bin_size == 0  # the reference's CPU path is the naive one
        got = orc.rasterize_meshes_naive(fv, first, count, nbr, size, blur, K, persp, clip, cull, cpu_order=True)
        assert torch.equal(got[0], outs[0])
        for a, b in zip(got[1:], outs[1:]):
            assert torch.equal(a, b), f"max diff {(a - b).abs().max().item()}"
    elif op == "rasterize_meshes_backward":
        fv, p2f, gz, gb, gd, persp, clip = args
        got = orc.rasterize_meshes_backward(fv, p2f, gz, gb, gd, persp, clip, cuda_semantics=False, acc64=False)
        _close(got, outs[0], atol=1e-6 * max(1.0, outs[0].abs().max().item()), rtol=1e-5)
    elif op == "_rasterize_meshes_coarse":
        fv, first, count, size, blur, bin_size, M = args
        got, overflow = orc.rasterize_meshes_coarse(fv, first, count, size, blur, bin_size, M)
        assert not overflow
        assert torch.equal(U.sort_bins(got), U.sort_bins(outs[0]))
    elif op == "rasterize_points":
        pts, first, count, size, radius, K, bin_size, M = args
        got = orc.rasterize_points_naive(pts, first, count, size, radius, K)
        assert torch.equal(got[0], outs[0].to(torch.int32))
        assert torch.equal(got[1], outs[1])
        _close(got[2], outs[2], atol=1e-6)
    elif op == "rasterize_points_backward":
        pts, idxs, gz, gd = args
        got = orc.rasterize_points_backward(pts, idxs, gz, gd, acc64=False)
        _close(got, outs[0], atol=2e-6 * max(1.0, outs[0].abs().max().item()), rtol=1e-5)
    elif op == "_rasterize_points_coarse":
        pts, first, count, size, radius, bin_size, M = args
        got, overflow = orc.rasterize_points_coarse(pts, first, count, size, radius, bin_size, M)
        assert not overflow
        assert torch.equal(U.sort_bins(got), U.sort_bins(outs[0]))
    elif op.startswith("accum_") and not op.endswith("_backward"):
        got = orc.composite_forward(op[len("accum_"):], *args)
        _close(got, outs[0], atol=2e-7)  # <= 1 ulp: CUDA vs CPU product order (alpha_composite.cu:64)
    elif op.endswith("_backward"):
        gf, ga = orc.composite_backward(op[len("accum_"):-len("_backward")], *args)
        _close(gf, outs[0], atol=1e-6, rtol=1e-6)
        _close(ga, outs[1], atol=1e-6, rtol=1e-6)
    else:
        pytest.fail(f"unhandled operator {op}")


def test_replay_covers_the_golden_bearing_reference_tests():
    tests = {t.split(".")[-1] for _, t, _, _ in CALLS}
    for needed in ("test_simple_cpu_naive", "test_coarse_cpu", "test_python_vs_cpp_perspective_correct",
                   "test_python_vs_cpp_bary_clip", "test_naive_simple_cpu", "test_cpu_variable_radius",
                   "test_cpu_behind_camera", "test_cpu"):
        assert needed in tests
