"""pytorch3d_amd/_aux_ops.py (the four small `_C` operators above the rasterization boundary that the shim provides as
torch formulations) against the reference's own CPU kernels (oracle/_ref/p3d_ref_cpu.so, build container only)."""
import pytest
import torch

import _util as U
from oracle import oracle as orc


def _ref():
    m = orc.ref_module()
    if m is None or not hasattr(m, "face_areas_normals_forward"):
        pytest.skip("oracle/_ref/p3d_ref_cpu.so not built")
    return m


def test_face_areas_normals_match_the_reference_cpu_kernels_including_its_gradient():
    ref = _ref()
    from pytorch3d_amd import _aux_ops as A

    v, f = U.ico_sphere(2)
    gen = torch.Generator().manual_seed(0)
    v = v + 0.05 * torch.randn(v.shape, generator=gen)
    a, n = ref.face_areas_normals_forward(v, f)
    a2, n2 = A.face_areas_normals_forward(v, f)
    assert torch.allclose(a, a2, atol=1e-6) and torch.allclose(n, n2, atol=1e-6)
    ga, gn = torch.randn(f.shape[0], generator=gen), torch.randn(f.shape[0], 3, generator=gen)
    want = ref.face_areas_normals_backward(ga, gn, v, f)
    got = A.face_areas_normals_backward(ga, gn, v, f)
    # includes the reference's c_x-for-c_y term in d / d(v1.z) (face_areas_normals.cu:183-184): a drop-in returns what the
    # reference returns; the analytic derivative differs from it by up to 17 on this input
    assert torch.allclose(got, want, atol=2e-5 * float(want.abs().max()))


def test_packed_padded_round_trip_matches_the_reference_cpu_kernels():
    ref = _ref()
    from pytorch3d_amd import _aux_ops as A

    gen = torch.Generator().manual_seed(1)
    x = torch.randn(20, 4, generator=gen)
    first = torch.tensor([0, 5, 5, 12])
    p, p2 = ref.packed_to_padded(x, first, 9), A.packed_to_padded(x, first, 9)
    assert torch.equal(p, p2)
    q, q2 = ref.padded_to_packed(p, first, 20), A.padded_to_packed(p, first, 20)
    assert torch.equal(q, q2) and torch.equal(q2, x)
