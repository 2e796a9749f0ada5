"""CPU suite, part 3: the C-ABI library and the host logic (no GPU, no compute calls).

* libp3d_amd.so loads and exports every function include/p3d_amd.h declares (and nothing in the
  header is missing from the ctypes table);
* entry points that can answer without a device (version, error strings, workspace sizing, argument
  validation that precedes any launch) behave;
* the `pytorch3d._C` surface mirrors the reference's pybind names / error behaviour and REFUSES CPU
  tensors (no fallback);
* host-side heuristics of the L2 mirror (bin size, max bins) follow the reference.
"""
import ctypes
import os
import re
import subprocess

import pytest
import torch

import _util as U

HEADER = os.path.join(U.ROOT, "include", "p3d_amd.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(p3d_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_operator_surface():
    names = _declared()
    for want in ("p3d_rasterize_meshes", "p3d_rasterize_meshes_naive", "p3d_rasterize_meshes_coarse",
                 "p3d_rasterize_meshes_fine", "p3d_rasterize_meshes_backward", "p3d_rasterize_points",
                 "p3d_rasterize_points_naive", "p3d_rasterize_points_coarse", "p3d_rasterize_points_fine",
                 "p3d_rasterize_points_backward", "p3d_composite_forward", "p3d_composite_backward", "p3d_composite_forward_strided",
                 "p3d_composite_backward_strided",
                 "p3d_interp_face_attrs_forward", "p3d_interp_face_attrs_backward", "p3d_sigmoid_alpha_blend_forward",
                 "p3d_sigmoid_alpha_blend_backward", "p3d_softmax_rgb_blend_forward", "p3d_softmax_rgb_blend_backward",
                 "p3d_gather_face_verts", "p3d_scatter_face_grads", "p3d_clip_faces_plan", "p3d_clip_faces_emit",
                 "p3d_clip_faces_backward", "p3d_convert_clipped_forward", "p3d_convert_clipped_backward"):
        assert want in names


def test_library_exports_every_declared_symbol():
    from pytorch3d_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "run `python -m pytorch3d_amd.build` (hipcc --offload-arch=gfx950)"
    dyn = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (p3d_[a-z0-9_]+)", dyn))
    declared = set(_declared())
    assert declared <= exported, f"declared but not exported: {sorted(declared - exported)}"
    assert exported <= declared, f"exported but not declared in include/p3d_amd.h: {sorted(exported - declared)}"
    assert set(_lib.EXPORTED_SYMBOLS) == declared, "ctypes table out of sync with the header"


def test_library_contains_gfx950_code_object():
    from pytorch3d_amd import _lib

    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"mesh_raster_kernel" in blob


def test_library_loads_and_answers_without_a_device():
    from pytorch3d_amd import _lib

    lib = _lib.load()
    assert lib.p3d_abi_version() == 1
    assert b"150" in lib.p3d_error_string(-2) or b"K" in lib.p3d_error_string(-2)
    assert lib.p3d_error_string(0)
    # workspace sizing is pure host arithmetic
    a = lib.p3d_rasterize_meshes_workspace_bytes(10000, 4, 128, 128, 16, 1000)
    b = lib.p3d_rasterize_meshes_workspace_bytes(20000, 4, 128, 128, 16, 1000)
    assert 0 < a <= b
    assert lib.p3d_rasterize_meshes_workspace_bytes(10000, 4, 128, 128, 0, 0) == 0
    # short workspaces: the fixed arrays + the list entries asked for, never above the worst case; the needed-entries word
    # lies inside the fixed part, 8-byte aligned
    bench = (321_000, 64, 512, 512, 32, 64_200)
    worst = lib.p3d_rasterize_meshes_workspace_bytes(*bench)
    s0 = lib.p3d_rasterize_meshes_short_workspace_bytes(*bench, 0)
    s1 = lib.p3d_rasterize_meshes_short_workspace_bytes(*bench, 2_100_000)
    assert 0 < s0 < s1 < worst and s1 - s0 == pytest.approx(4 * 2_100_000, abs=512)
    assert worst > 1.2e9 and s1 < 12e6  # the bench batch: 1.3 GB -> 11 MB
    assert lib.p3d_rasterize_meshes_short_workspace_bytes(*bench, 1 << 40) <= worst
    at = lib.p3d_rasterize_meshes_workspace_need_offset(*bench)
    assert 0 < at < s0 and at % 8 == 0
    assert lib.p3d_rasterize_meshes_short_workspace_bytes(10000, 4, 128, 128, 0, 0, 100) == 0
    # round 5: the worst-case size counts the marks of the CUDA tie order in (one 64-bit lane mask per 8 x 8 sub-tile, taken off
    # the END of the workspace); the short size does not -- a caller of it adds them (_C._mesh_workspace(extra=...))
    marks = 64 * (512 // 8) * (512 // 8) * 8
    assert lib.p3d_rasterize_meshes_short_workspace_bytes(*bench, 1 << 40) + marks <= worst
    assert lib.p3d_rasterize_points_workspace_bytes(10000, 2, 64, 64, 8, 100) > 0
    # ... and a point list entry is (id, depth bits): 8 bytes
    pts = (1_000_000, 1, 512, 512, 32, 20_000)
    p0 = lib.p3d_rasterize_points_short_workspace_bytes(*pts, 0)
    p1 = lib.p3d_rasterize_points_short_workspace_bytes(*pts, 2_000_000)
    assert p1 - p0 == pytest.approx(8 * 2_000_000, abs=512) and p1 <= lib.p3d_rasterize_points_workspace_bytes(*pts)
    assert lib.p3d_rasterize_fine_workspace_bytes(2, 4, 4, 10) >= 2 * 16 * 10 * 4
    # validation that precedes any launch: K > 150, too many bins, null outputs
    null = ctypes.c_void_p(None)
    rc = lib.p3d_rasterize_meshes_naive(null, null, null, null, 0, 1, 8, 8, 0.0, 151, 0, 0, 0, null, null, null, null, null)
    assert rc == -2
    rc = lib.p3d_rasterize_meshes(null, null, null, null, 0, 1, 64, 64, 0.0, 4, 2, 10, 0, 0, 0, null, null, null, null,
                                  null, 0, null)
    assert rc in (-1, -3)
    rc = lib.p3d_rasterize_meshes_coarse(null, null, null, 0, 1, 64, 64, 0.0, 2, 10, null, null, 0, null)
    assert rc == -3
    # empty problems return OK before touching the device
    rc = lib.p3d_rasterize_meshes_naive(null, null, null, null, 0, 0, 8, 8, 0.0, 4, 0, 0, 0, null, null, null, null, null)
    assert rc == 0


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from pytorch3d_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libp3d_amd.so"))
    with pytest.raises(_lib.ExtensionMissing, match="no CPU/eager fallback"):
        _lib.load()


def test_operator_surface_names_match_the_reference_pybind_module():
    """pytorch3d/csrc/ext.cpp:38-73 (hot-path subset) + the constants read at import (ext.cpp:180-185)."""
    from pytorch3d_amd import _C, shim

    ref_names = ["rasterize_meshes", "rasterize_meshes_backward", "_rasterize_meshes_naive", "_rasterize_meshes_coarse",
                 "_rasterize_meshes_fine", "rasterize_points", "rasterize_points_backward", "_rasterize_points_naive",
                 "_rasterize_points_coarse", "_rasterize_points_fine", "accum_alphacomposite",
                 "accum_alphacomposite_backward", "accum_weightedsumnorm", "accum_weightedsumnorm_backward",
                 "accum_weightedsum", "accum_weightedsum_backward", "interp_face_attrs_forward",
                 "interp_face_attrs_backward", "sigmoid_alpha_blend", "sigmoid_alpha_blend_backward"]
    mod = shim.make_module()
    for n in ref_names:
        assert callable(getattr(_C, n)) and callable(getattr(mod, n))
    assert (mod.EPS, mod.MAX_INT, mod.MAX_UINT, mod.MAX_USHORT, mod.PULSAR_MAX_GRAD_SPHERES) == (
        1e-6, 2147483647, 4294967295, 65535, 128)
    with pytest.raises(NotImplementedError):
        mod.knn_points_idx(None)


def test_cpu_tensors_are_refused_not_emulated():
    from pytorch3d_amd import _C

    fv = torch.rand(4, 3, 3)
    z = torch.zeros(1, dtype=torch.int64)
    with pytest.raises(RuntimeError, match="GPU path only"):
        _C.rasterize_meshes(fv, z, z + 4, torch.full((4,), -1, dtype=torch.int64), (8, 8), 0.0, 2, 0, 0, False, False,
                            False)
    with pytest.raises(RuntimeError, match="GPU path only"):
        _C.accum_alphacomposite(torch.rand(3, 5), torch.rand(1, 2, 4, 4), torch.zeros(1, 2, 4, 4, dtype=torch.int64))
    with pytest.raises(RuntimeError, match="GPU path only"):
        _C.interp_face_attrs_forward(torch.zeros(5, dtype=torch.int64), torch.rand(5, 3), torch.rand(2, 3, 4))


def test_l2_mirror_heuristics_follow_the_reference():
    """rasterize_meshes.py:195-222 / rasterize_points.py:104-126."""
    import importlib

    rm = importlib.import_module("pytorch3d_amd.rasterize_meshes")

    assert rm.default_bin_size(64) == 8
    assert rm.default_bin_size(65) == 16
    assert rm.default_bin_size(256) == 16
    assert rm.default_bin_size(512) == 32
    assert rm.default_bin_size(1024) == 64
    assert rm.parse_image_size(32) == (32, 32)
    assert rm.parse_image_size((20, 48)) == (20, 48)
    with pytest.raises(ValueError):
        rm.parse_image_size((0, 4))

    class M:
        _F = 10

        def verts_packed(self):
            return torch.rand(6, 3)

        def faces_packed(self):
            return torch.tensor([[0, 1, 2], [3, 4, 5]])

        def mesh_to_faces_packed_first_idx(self):
            return torch.tensor([0])

        def num_faces_per_mesh(self):
            return torch.tensor([2])

    with pytest.raises(ValueError, match="bin_size too small"):
        rm.rasterize_meshes(M(), image_size=512, bin_size=8)  # tests/test_rasterize_meshes.py:461-466
    with pytest.raises(RuntimeError, match="GPU path only"):  # clipping is implemented: it reaches the GPU check
        rm.rasterize_meshes(M(), image_size=32, z_clip_value=0.1)


def test_packed_containers_match_reference_accessors():
    import pytorch3d_amd as p3d

    v = [torch.rand(5, 3), torch.rand(7, 3)]
    f = [torch.tensor([[0, 1, 2], [2, 3, 4]]), torch.tensor([[0, 5, 6]])]
    m = p3d.PackedMeshes(v, f)
    assert len(m) == 2
    assert m.verts_packed().shape == (12, 3)
    assert m.faces_packed().tolist() == [[0, 1, 2], [2, 3, 4], [5, 10, 11]]
    assert m.mesh_to_faces_packed_first_idx().tolist() == [0, 2]
    assert m.num_faces_per_mesh().tolist() == [2, 1]
    assert m._F == 2


def test_shading_host_logic_packs_broadcasts_and_refuses():
    """pytorch3d_amd/shading.py host side (no GPU): the (N, 25) parameter block of p3d_phong_shade_*, light-kind
    detection, (1,3)/(N,3) broadcasting, loud refusals."""
    from collections import namedtuple

    import pytorch3d_amd as p3d
    import pytorch3d_amd.shading as sh

    class Cam:
        def get_camera_center(self):
            return torch.tensor([[0.0, 0.0, -3.0]])

    M = sh.Materials(torch.ones(1, 3), torch.full((1, 3), 0.5), torch.full((1, 3), 0.25), torch.tensor([64.0]))
    point = sh.Lights(torch.tensor([[0.1, 0.2, 0.3]]), torch.rand(3, 3), torch.rand(1, 3), location=torch.rand(3, 3))
    params, kind = sh.pack_shade_params(point, Cam(), M, 3, torch.device("cpu"))
    assert params.shape == (3, sh.PARAM_FLOATS) and params.is_contiguous() and kind == sh.LIGHT_POINT
    assert torch.equal(params[:, 0:3], torch.tensor([[0.1, 0.2, 0.3]]).expand(3, 3))
    assert torch.equal(params[:, 3:6], point.diffuse_color) and torch.equal(params[:, 9:12], point.location)
    assert torch.equal(params[:, 21], torch.full((3,), 64.0)) and torch.equal(params[:, 22:25], torch.tensor([[0.0, 0.0, -3.0]]).expand(3, 3))
    direc = sh.Lights(torch.ones(1, 3), torch.ones(1, 3), torch.ones(1, 3), direction=torch.tensor([[0.0, 1.0, 0.0]]))
    assert sh.pack_shade_params(direc, Cam(), M, 2, torch.device("cpu"))[1] == sh.LIGHT_DIRECTIONAL
    amb = sh.Lights(torch.full((1, 3), 0.7), diffuse_color=torch.ones(1, 3))  # ambient-only: diffuse / specular ignored
    pa, ka = sh.pack_shade_params(amb, Cam(), M, 2, torch.device("cpu"))
    assert ka == sh.LIGHT_DIRECTIONAL and pa[:, 3:12].abs().max() == 0 and torch.all(pa[:, 0:3] == 0.7)
    with pytest.raises(ValueError, match="must have shape"):
        sh.pack_shade_params(point, Cam(), M, 2, torch.device("cpu"))  # (3,3) lights against a batch of 2
    # the packing is differentiable: a light / material / camera tensor that requires grad gets its gradient through it
    loc = torch.rand(3, 3, requires_grad=True)
    pg, _ = sh.pack_shade_params(point._replace(location=loc), Cam(), M, 3, torch.device("cpu"))
    pg[:, 9:12].sum().backward()
    assert torch.equal(loc.grad, torch.ones(3, 3))
    # no CPU emulation of the fused kernels
    Frag = namedtuple("Frag", "pix_to_face bary_coords")
    m = p3d.PackedMeshes([torch.rand(4, 3)], [torch.tensor([[0, 1, 2], [1, 2, 3]])])
    frag = Frag(torch.zeros(1, 2, 2, 1, dtype=torch.int64), torch.rand(1, 2, 2, 1, 3))
    with pytest.raises(RuntimeError, match="GPU path only"):
        p3d.phong_shading(m, frag, direc, Cam(), M, torch.rand(1, 2, 2, 1, 3))
    # the container's normals follow Meshes._compute_vertex_normals / mesh_face_areas_normals
    assert torch.allclose(m.verts_normals_packed().norm(dim=1), torch.ones(4), atol=1e-6)
    assert torch.allclose(m.faces_normals_packed().norm(dim=1), torch.ones(2), atol=1e-6)
    assert m.verts_packed_to_mesh_idx().tolist() == [0, 0, 0, 0]


def test_uv_sampling_host_logic_refuses_loudly():
    """pytorch3d_amd/textures.py host side (no GPU): argument checks, no CPU emulation."""
    from collections import namedtuple

    import pytorch3d_amd as p3d

    Frag = namedtuple("Frag", "pix_to_face bary_coords")
    frag = Frag(torch.zeros(1, 2, 2, 1, dtype=torch.int64), torch.rand(1, 2, 2, 1, 3))
    fu, maps = torch.rand(3, 3, 2), torch.rand(1, 4, 4, 3)
    with pytest.raises(NotImplementedError, match="padding_mode"):
        p3d.sample_textures_uv(frag, fu, maps, padding_mode="reflection")
    with pytest.raises(ValueError, match="sampling_mode"):
        p3d.sample_textures_uv(frag, fu, maps, sampling_mode="bicubic")
    with pytest.raises(RuntimeError, match="GPU path only"):
        p3d.sample_textures_uv(frag, fu, maps)


def test_shared_reciprocal_division_is_exact_and_fast_path_equals_plain_path():
    """csrc/p3d_geom.h, compiled for the host: (1) exact_div -- (float)((double)n * rd) -- equals IEEE n / d for every rd
    within one double ulp of 1/d (30M triples); (2) the fine kernel's fast path (rectangle prune + face_hit_rec on
    FaceRecs) produces bit-identical rasterizations to the plain reference-order path on random soups, with and without
    blur, all flag combinations, including degenerate and tiny faces."""
    import ctypes

    import _util as U

    hg = U.hostgeom()
    hg.hg_exact_div_check.restype = ctypes.c_int64
    hg.hg_exact_div_check.argtypes = [ctypes.c_int64, ctypes.c_uint64]
    assert hg.hg_exact_div_check(10_000_000, 12345) == 0
    gen = torch.Generator().manual_seed(5)
    for case, (blur, persp, clip, cull) in enumerate([(0.0, False, False, False), (1e-4, True, True, False),
                                                     (9.2e-4, True, True, False), (0.01, True, False, True),
                                                     (0.003, False, True, False)]):
        F = 160
        fv = U.triangle_soup(F, gen, size=0.6 if case % 2 else 0.15, behind_every=13)
        fv[5::17, 2] = fv[5::17, 1] + 1e-5  # slivers with a near-degenerate edge
        fv[3::29, 1, :2] = fv[3::29, 0, :2]  # exactly degenerate edge
        fv[7::31] *= torch.tensor([1e-3, 1e-3, 1.0])  # tiny faces near the image centre
        first, count = U.split_counts(F, 2)
        nbr = torch.full((F,), -1, dtype=torch.int64)
        for size in ((40, 40), (24, 56)):
            plain = U.hg_rasterize_meshes(fv, first, count, nbr, size, blur, 8, persp, clip, cull, use_mem=0)
            fast = U.hg_rasterize_meshes(fv, first, count, nbr, size, blur, 8, persp, clip, cull, use_mem=2)
            for a, b in zip(plain, fast):
                assert torch.equal(a, b), (case, size)


def test_pixel_masks_equal_the_per_pixel_bbox_test():
    """csrc/p3d_geom.h: range_mask16 / block_mask_8x8 -- the fine rasterizer takes a face's bounding-box test once per
    (face, tile column) and (face, tile row) and hands the pixels inside to a visit as a 64-bit lane mask; the reference
    tests every pixel (rasterize_meshes.cu:94-97, strict comparisons).  Host build of the header: 128M (pixel, box) pairs on
    random image sizes, partial tiles, edges exactly on pixel centres, NaN edges -- the mask bit equals the per-pixel test."""
    import ctypes

    import _util as U

    hg = U.hostgeom()
    hg.hg_pixel_mask_check.restype = ctypes.c_int64
    hg.hg_pixel_mask_check.argtypes = [ctypes.c_int64, ctypes.c_uint64]
    assert hg.hg_pixel_mask_check(500_000, 3) == 0


def test_point_rasterizer_host_logic():
    """pytorch3d_amd/rasterize_points.py: radius forms, the reference's bin-size heuristic (rasterize_points.py:104-113, no
    "<= 64 -> 8" special case as meshes have) and its error; the kernels themselves refuse CPU tensors."""
    import importlib

    import pytorch3d_amd as p3d

    rp = importlib.import_module("pytorch3d_amd.rasterize_points")  # the package re-exports the function under this name

    assert [rp.default_bin_size(s) for s in (16, 64, 256, 257, 512, 1024)] == [16, 16, 16, 32, 32, 64]
    pts = [torch.rand(5, 3), torch.rand(3, 3)]
    pc = p3d.PackedPointclouds(pts)
    r = rp.radius_per_packed_point(0.25, pc)
    assert r.shape == (8,) and r.dtype == torch.float32 and bool((r == 0.25).all())
    packed = torch.arange(8, dtype=torch.float32)
    assert torch.equal(rp.radius_per_packed_point(packed, pc), packed)
    padded = torch.arange(10, dtype=torch.float32).reshape(2, 5)
    assert torch.equal(rp.radius_per_packed_point(padded, pc), torch.tensor([0., 1., 2., 3., 4., 5., 6., 7.]))
    one = p3d.PackedPointclouds([torch.rand(4, 3)])
    assert torch.equal(rp.radius_per_packed_point([1.0, 2.0, 3.0, 4.0], one), torch.tensor([1., 2., 3., 4.]))
    with pytest.raises(ValueError, match="shape"):
        rp.radius_per_packed_point(torch.rand(2, 4), pc)
    with pytest.raises(ValueError, match="float, list, tuple or tensor"):
        rp.radius_per_packed_point("0.1", pc)
    with pytest.raises(ValueError, match="bin_size too small"):
        p3d.rasterize_points(pc, image_size=512, bin_size=8)
    with pytest.raises(RuntimeError, match="GPU path only"):
        p3d.rasterize_points(pc, image_size=32, radius=0.1)


def test_bench_jobs_partition_and_fast_path_on_the_cow():
    """(1) bench.py --jobs (BASELINE configs[4]): 512 jobs = 8 sub-batches of 64 dealt to 1 / 2 / 4 / 8 ranks, every sub-batch
    exactly once.  (2) The fine kernel's fast path (csrc/p3d_geom.h compiled for the host: rectangle prune + FaceRec
    evaluation) on REAL geometry: the reference's cow at 96x96, K=8, blur 1e-4 -- bit-identical to the plain reference-order
    path and to the C oracle."""
    import importlib.util

    import numpy as np

    import _util as U
    from oracle import oracle as orc

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(U.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for world in (1, 2, 4, 8):
        got = sorted(s for r in range(world) for s in bench.sub_batches_of_rank(512, 64, r, world))
        assert got == list(range(8))
        assert all(len(bench.sub_batches_of_rank(512, 64, r, world)) == 8 // world for r in range(world))
    # an uneven deal (round 6): 8 sub-batches over 3 ranks = 3 + 3 + 2, every sub-batch exactly once; rank 0 never has fewer than another
    deal = [bench.sub_batches_of_rank(512, 64, r, 3) for r in range(3)]
    assert [len(d) for d in deal] == [3, 3, 2] and sorted(s for d in deal for s in d) == list(range(8))
    with pytest.raises(SystemExit):
        bench.sub_batches_of_rank(512, 64, 0, 9)  # a rank would have nothing to run
    with pytest.raises(SystemExit):
        bench.sub_batches_of_rank(500, 64, 0, 1)

    g = np.load(os.path.join(U.GOLDEN, "cow_ref.npz"))
    fv = torch.from_numpy(g["verts_ndc"])[torch.from_numpy(g["faces"]).long()].contiguous()
    F = fv.shape[0]
    first = torch.zeros(1, dtype=torch.int64)
    count = torch.tensor([F], dtype=torch.int64)
    nbr = torch.full((F,), -1, dtype=torch.int64)
    size, blur, K = (96, 96), 1e-4, 8
    plain = U.hg_rasterize_meshes(fv, first, count, nbr, size, blur, K, True, True, False, use_mem=0)
    fast = U.hg_rasterize_meshes(fv, first, count, nbr, size, blur, K, True, True, False, use_mem=2)
    ref = orc.rasterize_meshes_naive(fv, first, count, nbr, size, blur, K, True, True, False)
    for a, b, c in zip(plain, fast, ref):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert int((ref[0] >= 0).sum()) > 1000


def test_pair_register_queue_scheme_matches_the_register_queue():
    """csrc/topk.h: TopKPairs (entries in 64-bit register pairs, insertion = two masked pair moves per entry; the queue of
    the K = 4, 8, 16 mesh kernels and of the common point capacities since round 3) against TopKReg on 200k random
    operation sequences with depth ties, repeated indices, find and erase.  On the host the masked moves are plain ifs:
    this pins the scheme; the v_pk_mov_b32 / exec sequence itself is what the whole GPU suite runs."""
    import ctypes

    hg = U.hostgeom()
    hg.hg_queue_pairs_check.restype = ctypes.c_int64
    hg.hg_queue_pairs_check.argtypes = [ctypes.c_int64, ctypes.c_uint64]
    assert hg.hg_queue_pairs_check(200_000, 99) == 0
    # the payload-free form of the point queues, with and without the 64-bit key compare
    hg.hg_queue_pairs0_check.restype = ctypes.c_int64
    hg.hg_queue_pairs0_check.argtypes = [ctypes.c_int64, ctypes.c_uint64]
    assert hg.hg_queue_pairs0_check(200_000, 7) == 0


def test_no_kernel_of_the_library_spills_vgprs():
    """pytorch3d_amd/build.py records the compiler's per-kernel resource usage next to the library and refuses a build in which
    a kernel spills VGPRs: in round 4 every kernel of raster_mesh.hip that did lost queue entries on the GPU, bit-exact again as
    soon as it fitted its registers (profiles/r04/spill_miscompile.md).  Scratch itself is fine where it is the design (the
    private-memory queue TopKMem for K > 64 / > 100; the replay of the reference's 150-entry array in the *_cuda_order kernels)."""
    import json

    from pytorch3d_amd import build as B

    path = B.LIB + ".resources.json"
    assert os.path.exists(path), "build the library with `python -m pytorch3d_amd.build` (conftest does)"
    res = json.load(open(path))
    assert len(res) > 150
    spilled = {B.demangle(k): v["vgpr_spill"] for k, v in res.items() if v["vgpr_spill"] != 0}
    assert not spilled, spilled
    scratch = sorted(B.demangle(k) for k, v in res.items() if v["scratch"] > 0)
    assert all("TopKMem" in k or "_cuda_order_kernel" in k for k in scratch), scratch
    # round 5: registers kept in AGPRs only in kernels that name the GPU test which fills every register row
    agpr = sorted(B.demangle(k) for k, v in res.items() if v["agprs"] > 0)
    assert all(any(t + "(" in k for t in B.AGPR_KERNELS_TESTED) for k in agpr), agpr


def test_short_workspace_sizing_follows_a_running_maximum_and_reports_while_tight():
    """pytorch3d_amd/_C.py: _Need: the list size of a call shape is the running maximum (slow decay) of what its
    calls reported, and a call whose last report used more than 80 % of its lists asks for a report again at once -- not the last
    report alone, read back every 8th call (up to seven calls in a row could then run the naive stand-by on a growing scene)."""
    from pytorch3d_amd import _C

    class Done:
        def query(self):
            return True

    need = _C._Need()

    def report(v, capacity):
        need.pinned = torch.tensor([v], dtype=torch.int64)
        need.event = Done()
        need.capacity = capacity
        need.collect()

    report(1000, 5000)
    assert need.entries == 1000 and need.last == 1000 and not need.tight()
    report(4000, 5000)  # growth is taken at once
    assert need.entries == 4000 and not need.tight()
    report(4100, 5000)
    assert need.entries == 4100 and need.tight()  # 82 % of the lists: the next call reports again whatever its number
    report(2000, 5000)  # a smaller report pulls the size down slowly
    assert need.entries == 4100 - (4100 - 2000) // 8 and not need.tight()
    report(9000, 5000)  # overflow (the stand-by kernel ran): tight, and the size jumps
    assert need.entries == 9000 and need.tight()
    need.calls, need.event = 100, None  # far from the first four calls, not a multiple of 8
    need.calls = 101
    asked = []

    class FakeWs:
        def __getitem__(self, sl):
            asked.append(sl)
            raise RuntimeError("stop here")  # the copy itself needs a device

    try:
        need.report_later(FakeWs(), 0, 5000)
    except RuntimeError:
        pass
    assert asked, "a tight shape did not ask for a report"
    report(1000, 50000)
    need.calls = 101
    asked.clear()
    need.report_later(FakeWs(), 0, 50000)
    assert not asked, "a roomy shape asked for a report on a call that is not its eighth"


def test_bench_jobs_mode_deals_every_sub_batch_to_exactly_one_rank():
    """BASELINE configs[4]: 512 jobs = 8 sub-batches of 64 (generator seeds 0..7); on G in
    {1, 2, 4, 8} GPUs rank r runs sub-batches r, r + G, ...: every seed exactly once, the same number per rank.  A G that does
    not divide deals unevenly (round 6: the gather pads the shards to the largest, bench.py: `own`); a G with an idle rank is refused."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(U.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for world in (1, 2, 4, 8):
        dealt = [bench.sub_batches_of_rank(512, 64, r, world) for r in range(world)]
        assert sorted(s for d in dealt for s in d) == list(range(8))
        assert len({len(d) for d in dealt}) == 1 and len(dealt[0]) == 8 // world
        assert all(d == sorted(d) and d[0] == r for r, d in enumerate(dealt))
    for world in (3, 5):
        dealt = [bench.sub_batches_of_rank(512, 64, r, world) for r in range(world)]
        assert sorted(s for d in dealt for s in d) == list(range(8)) and max(len(d) for d in dealt) == len(dealt[0]) == -(-8 // world)
    with pytest.raises(SystemExit):
        bench.sub_batches_of_rank(512, 64, 0, 16)
    with pytest.raises(SystemExit):
        bench.sub_batches_of_rank(500, 64, 0, 2)


def test_pybind_flavour_builds_and_exports_the_operator_set():
    """pytorch3d_amd/csrc/bind.cpp (INTEGRATION.md section B compiled): the torch extension builds against include/p3d_amd.h, links
    libp3d_amd.so and exports every hot-path operator of `pytorch3d._C` under the reference's names (no compute call without a GPU)."""
    from pytorch3d_amd import _C, build_bind

    try:
        mod = build_bind.load()
    except Exception as e:  # noqa: BLE001
        pytest.skip("no host toolchain for the pybind flavour here: %r" % (e,))
    missing = [n for n in _C.HOT_PATH_EXPORTS if not callable(getattr(mod, n, None))]
    assert not missing, missing
    assert mod.MAX_INT == 2147483647 and mod.__p3d_amd_flavour__ == "pybind"
    with pytest.raises(RuntimeError, match="GPU tensor"):  # CPU tensors are refused, as in the ctypes flavour
        z = torch.zeros(1, dtype=torch.int64)
        mod.rasterize_meshes(torch.rand(4, 3, 3), z, z + 4, torch.full((4,), -1, dtype=torch.int64), (8, 8), 0.0, 2, 0, 0, False, False, False)
