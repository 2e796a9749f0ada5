"""CPU suite, part 5 (build container only: needs the read-only reference checkout): the UNMODIFIED reference
package -- Meshes, cameras, MeshRasterizer, PointsRasterizer, AlphaCompositor, sigmoid_alpha_blend -- is imported
with `pytorch3d._C` provided by pytorch3d_amd.shim, and driven until it calls our operators.  There is no GPU
here, so the calls must arrive at pytorch3d_amd._C with the reference's positional signature (a mismatch would be
a TypeError) and be refused with our "GPU path only" error -- which proves the plumbing and that no CPU fallback
hides behind the boundary.  Runs in a subprocess so that the shim does not leak into the other tests.
"""
import os
import subprocess
import sys
import textwrap

import pytest

REFERENCE = os.environ.get("P3D_REFERENCE_ROOT", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "pytorch3d", "renderer")),
                                reason="reference checkout not present (GPU box)")

SCRIPT = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    import torch
    import pytorch3d_amd.shim as shim
    mod = shim.install(%r)
    import pytorch3d
    from pytorch3d import _C
    assert _C is mod and getattr(_C, "__p3d_amd__", False)
    from pytorch3d.structures import Meshes, Pointclouds
    from pytorch3d.renderer import (FoVPerspectiveCameras, MeshRasterizer, RasterizationSettings, PointsRasterizer,
                                    PointsRasterizationSettings, look_at_view_transform)
    from pytorch3d.renderer.compositing import alpha_composite
    from pytorch3d.renderer.blending import sigmoid_alpha_blend, BlendParams
    from pytorch3d.utils import ico_sphere

    def refused(fn):
        try:
            fn()
        except RuntimeError as e:
            assert "GPU path only" in str(e), e
            return True
        raise AssertionError("the call did not reach pytorch3d_amd._C")

    R, T = look_at_view_transform(2.7, 0, 0)
    cams = FoVPerspectiveCameras(R=R, T=T)
    meshes = ico_sphere(1)
    # bin_size=None on CPU tensors -> the reference picks the naive path; force the binned signature as well
    for bs in (0, 8):
        rast = MeshRasterizer(cameras=cams, raster_settings=RasterizationSettings(image_size=32, blur_radius=1e-4,
                              faces_per_pixel=4, bin_size=bs, max_faces_per_bin=100))
        assert refused(lambda: rast(meshes))
    pts = Pointclouds(points=[torch.rand(50, 3) + torch.tensor([0.0, 0.0, 2.0])])
    prast = PointsRasterizer(cameras=cams, raster_settings=PointsRasterizationSettings(image_size=16, radius=0.1,
                             points_per_pixel=3, bin_size=0))
    assert refused(lambda: prast(pts))
    assert refused(lambda: alpha_composite(torch.zeros(1, 2, 4, 4, dtype=torch.int64), torch.rand(1, 2, 4, 4),
                                           torch.rand(3, 5)))
    from collections import namedtuple
    F = namedtuple("F", "pix_to_face dists")
    assert refused(lambda: sigmoid_alpha_blend(torch.rand(1, 4, 4, 2, 3), F(torch.zeros(1, 4, 4, 2, dtype=torch.int64),
                                                                            torch.rand(1, 4, 4, 2)), BlendParams()))
    # operators outside the hot path are stubs that say so
    try:
        _C.knn_points_idx(None)
    except NotImplementedError as e:
        assert "outside the rasterization hot path" in str(e)
    print("SHIM-OK")
""")


def test_unmodified_reference_reaches_our_operators_through_the_shim():
    out = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, REFERENCE)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "SHIM-OK" in out.stdout, out.stderr[-3000:]
