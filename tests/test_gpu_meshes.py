"""GPU parity: mesh rasterization (coarse / fine / naive / backward) vs the oracle and vs the
reference's own CPU build, through the C ABI (pytorch3d_amd._C -> libp3d_amd.so).

Tolerances (north_star): pix_to_face and bin indices bit-exact; zbuf / bary / dists within 1e-5
(they are in fact compared bit-exact against the CUDA-order oracle); gradients within the
reference's own tolerances (tests/test_rasterize_meshes.py:317-319,594: rtol 2e-3..5e-3).
"""
import pytest
import torch

import _util as U
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _run_ours(fv, first, count, nbr, size, blur, K, bin_size, M, persp, clip, cull):
    from pytorch3d_amd import _C

    d = _dev()
    out = _C.rasterize_meshes(fv.to(d), first.to(d), count.to(d), nbr.to(d), size, blur, K, bin_size, M, persp, clip,
                              cull)
    torch.cuda.synchronize()
    return [o.cpu() for o in out]


def _assert_fwd_equal(ours, ref, exact_floats=True, tag=""):
    assert torch.equal(ours[0], ref[0]), f"pix_to_face differs {tag}: {(ours[0] != ref[0]).sum().item()} elements"
    for name, a, b in zip(("zbuf", "bary", "dists"), ours[1:], ref[1:]):
        if exact_floats:
            assert torch.equal(a, b), f"{name} not bit-exact {tag}: max diff {(a - b).abs().max().item()}"
        else:
            assert torch.allclose(a, b, atol=1e-5, rtol=0), f"{name} beyond 1e-5 {tag}: {(a - b).abs().max().item()}"


@pytest.mark.parametrize("size", [(32, 32), (20, 48), (37, 23), (64, 64)])
@pytest.mark.parametrize("blur", [0.0, 0.01])
@pytest.mark.parametrize("persp,clip,cull", [(False, False, False), (True, False, False), (True, True, False),
                                             (False, True, True), (True, True, True)])
def test_naive_and_binned_vs_oracle(size, blur, persp, clip, cull):
    gen = torch.Generator().manual_seed(hash((size, blur, persp, clip, cull)) % 1000)
    F, N, K = 120, 3, 4
    fv = U.triangle_soup(F, gen, behind_every=11)
    first, count = U.split_counts(F, N)
    nbr = torch.full((F,), -1, dtype=torch.int64)
    ref = orc.rasterize_meshes_naive(fv, first, count, nbr, size, blur, K, persp, clip, cull)
    naive = _run_ours(fv, first, count, nbr, size, blur, K, 0, 0, persp, clip, cull)
    _assert_fwd_equal(naive, ref, tag="naive")
    for bin_size in (8, 16, 5):
        if 1 + (max(size) - 1) // bin_size >= 22:
            continue
        binned = _run_ours(fv, first, count, nbr, size, blur, K, bin_size, 200, persp, clip, cull)
        _assert_fwd_equal(binned, ref, tag=f"bin_size={bin_size}")


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 15, 16, 17, 20, 23, 24, 25, 31, 32, 33, 40, 41, 47, 48, 49, 50, 56, 63, 64, 65, 100, 150])
def test_all_queue_capacities(K):
    """Every queue of the launcher (raster_mesh.hip: launch_mesh_raster): register queues with payload up to 16, queues
    without payload (distance / barycentrics recomputed at the store) for 17..64 at their exact capacities and at the
    next one above K, the private-memory queue beyond -- with and without the perspective + clip instantiation
    (64-bit key compares), binned and naive."""
    gen = torch.Generator().manual_seed(K)
    F = 300
    fv = U.triangle_soup(F, gen, size=1.0)
    first, count = U.split_counts(F, 2)
    nbr = torch.full((F,), -1, dtype=torch.int64)
    for persp, clip in ((True, True), (True, False), (False, False)):
        ref = orc.rasterize_meshes_naive(fv, first, count, nbr, (24, 24), 0.005, K, persp, clip, False)
        ours = _run_ours(fv, first, count, nbr, (24, 24), 0.005, K, 8, 300, persp, clip, False)
        _assert_fwd_equal(ours, ref, tag=f"K={K} persp={persp} clip={clip}")
    # A soup of frame-sized triangles: ~100 faces over every pixel, so every queue up to 64 entries OVERFLOWS (asserted) -- the
    # regime in which round 4's kernels with VGPR spills lost entries while the sparse soup above still passed
    # (profiles/r04/spill_miscompile.md); two chunks of faces in the naive path; a non-square image with partial tiles.
    big = U.triangle_soup(260, gen, size=4.0)
    bfirst, bcount = U.split_counts(260, 1)
    bnbr = torch.full((260,), -1, dtype=torch.int64)
    for persp, clip in ((True, True), (False, False)):
        ref = orc.rasterize_meshes_naive(big, bfirst, bcount, bnbr, (20, 37), 0.02, K, persp, clip, False)
        assert K > 64 or int((ref[0][..., K - 1] >= 0).sum()) > 0, "the soup does not fill a queue of this length"
        for bin_size in (0, 8):
            ours = _run_ours(big, bfirst, bcount, bnbr, (20, 37), 0.02, K, bin_size, 300, persp, clip, False)
            _assert_fwd_equal(ours, ref, tag=f"K={K} persp={persp} clip={clip} 20x37 bin={bin_size}")


@pytest.mark.parametrize("K", [49, 64, 100, 150])
@pytest.mark.parametrize("persp,clip", [(True, True), (False, False)])
def test_k_above_the_register_queues_on_a_scene_with_sparse_and_dense_tiles(K, persp, clip):
    """raster_mesh.hip, K > 48 (the private-memory queue; round 5 tried a register first pass + redo of the tiles with a full queue
    on this test: profiles/r05/exp_two_pass_k49.md, dropped).  A scene with BOTH kinds of tiles: two
    small meshes seen with SoftRas blur (a covered pixel holds ~10-30 faces: first-pass tiles) and, in one corner, a stack of 60
    coincident-footprint triangles at different depths (every pixel under it overflows 32: redone tiles) -- against the oracle,
    binned; images with ragged tiles; with the reference's neighbour rule in play (GENERAL nest) in a third mesh."""
    gen = torch.Generator().manual_seed(4000 + K)
    verts, faces = U.hetero_batch(2, seed=11, fmin=300, fmax=700)
    from pytorch3d_amd import PackedMeshes

    m = PackedMeshes(verts, faces)
    fv = m.verts_packed()[m.faces_packed()].contiguous()
    stack = torch.tensor([[-0.95, -0.95, 1.0], [-0.45, -0.95, 1.0], [-0.7, -0.5, 1.0]]).repeat(60, 1, 1)
    stack[:, :, 2] += torch.rand(60, 1, generator=gen) * 2.0
    stack[:, :, :2] += (torch.rand(60, 3, 2, generator=gen) - 0.5) * 0.02
    soup = U.triangle_soup(40, gen)
    fv = torch.cat([fv, stack, soup], 0).contiguous()
    first = torch.cat([m.mesh_to_faces_packed_first_idx(), torch.tensor([int(m.faces_packed().shape[0]) + 60])])
    count = torch.cat([m.num_faces_per_mesh(), torch.tensor([40])])
    count[1] += 60  # the stack belongs to the second mesh
    nbr = torch.full((fv.shape[0],), -1, dtype=torch.int64)
    s0 = int(first[2])
    for a in range(s0, s0 + 40, 4):
        nbr[a], nbr[a + 1] = a + 1, a
    size, blur = (72, 100), 2e-3
    ref = orc.rasterize_meshes_naive(fv, first, count, nbr, size, blur, K, persp, clip, False)
    per_pixel = (ref[0] >= 0).sum(-1)
    assert int((per_pixel > 32).sum()) > 0 and int(((per_pixel > 0) & (per_pixel < 32)).sum()) > 0, "the scene should hold both kinds of pixels"
    for bin_size in (16, 32):
        ours = _run_ours(fv, first, count, nbr, size, blur, K, bin_size, 2000, persp, clip, False)
        _assert_fwd_equal(ours, ref, tag=f"K={K} bin={bin_size}")


@pytest.mark.parametrize("K", [1, 4, 8, 12, 50])
def test_needle_faces_with_a_degenerate_edge(K):
    """Faces with ONE edge shorter than 1e-4 (squared length <= 1e-8: the reference takes the distance to that edge as the distance
    to its end point, geometry_utils.cuh:345) but an area well above 1e-8 -- needles.  The perspective + clip kernels evaluate
    ordinary faces without that alternative and hand chunks that hold such a face to their general nest (raster_mesh.hip:
    stage_chunk, p3d_geom.h: face_rec_degenerate): needles among ordinary faces, in every position of a chunk, several chunks per
    tile, all flag combinations, against the oracle."""
    gen = torch.Generator().manual_seed(700 + K)
    F = 620
    fv = U.triangle_soup(F, gen, size=0.6)
    needles = torch.randperm(F, generator=gen)[:90]
    for j, i in enumerate(needles.tolist()):
        a, b = (0, 1) if j % 3 == 0 else ((1, 2) if j % 3 == 1 else (2, 0))
        step = (torch.rand(2, generator=gen) - 0.5) * (3e-4 if j % 2 else 2e-5)  # both sides of the 1e-4 threshold
        fv[i, b, :2] = fv[i, a, :2] + step
    first, count = U.split_counts(F, 2)
    nbr = torch.full((F,), -1, dtype=torch.int64)
    for persp, clip in ((True, True), (False, False), (True, False)):
        ref = orc.rasterize_meshes_naive(fv, first, count, nbr, (40, 56), 0.004, K, persp, clip, False)
        assert int(torch.isin(ref[0], needles).sum()) > 0, "no needle is visible"
        for bin_size in (0, 16, 32):
            ours = _run_ours(fv, first, count, nbr, (40, 56), 0.004, K, bin_size, 700, persp, clip, False)
            _assert_fwd_equal(ours, ref, tag=f"K={K} persp={persp} clip={clip} bin={bin_size}")


def test_k_too_large_raises():
    from pytorch3d_amd import _C

    d = _dev()
    fv = torch.rand(4, 3, 3, device=d)
    z = torch.zeros(1, dtype=torch.int64, device=d)
    with pytest.raises(RuntimeError, match="Must have points_per_pixel <= 150"):
        _C.rasterize_meshes(fv, z, z + 4, torch.full((4,), -1, dtype=torch.int64, device=d), (8, 8), 0.0, 151, 0, 0,
                            False, False, False)


def test_too_many_bins_raises():
    from pytorch3d_amd import _C

    d = _dev()
    fv = torch.rand(4, 3, 3, device=d)
    z = torch.zeros(1, dtype=torch.int64, device=d)
    with pytest.raises(RuntimeError, match="that's too many"):
        _C._rasterize_meshes_coarse(fv, z, z + 4, (64, 64), 0.0, 2, 10)


def test_order_of_ties():
    """Faces at exactly the same depth are ordered by face index (tests/test_rasterize_meshes.py:1165-1185)."""
    F = 12
    tri = torch.tensor([[-0.8, -0.8, 1.0], [0.8, -0.8, 1.0], [0.0, 0.8, 1.0]])
    fv = tri[None].repeat(F, 1, 1)
    first = torch.tensor([0])
    count = torch.tensor([F])
    nbr = torch.full((F,), -1, dtype=torch.int64)
    for K in (8, 100):
        for bin_size in (0, 8):
            ours = _run_ours(fv, first, count, nbr, (16, 16), 0.0, K, bin_size, 50, False, False, False)
            hit = ours[0][0, 8, 8]
            expect = torch.cat([torch.arange(min(F, K)), torch.full((max(K - F, 0),), -1)])
            assert torch.equal(hit[:K], expect[:K])


def test_clipped_neighbor_rule():
    gen = torch.Generator().manual_seed(5)
    F = 40
    fv = U.triangle_soup(F, gen)
    nbr = torch.full((F,), -1, dtype=torch.int64)
    for a in range(0, F, 4):
        nbr[a], nbr[a + 1] = a + 1, a
        fv[a + 1] = fv[a] + torch.randn(3, 3, generator=gen) * 0.05
    first, count = U.split_counts(F, 1)
    for K in (2, 5, 17, 20, 30, 32, 64):  # 17..64: queues without payload -- the rule's distance test recomputes the queued half
        ref = orc.rasterize_meshes_naive(fv, first, count, nbr, (24, 24), 0.01, K, True, True, False)
        for bin_size in (0, 8):
            ours = _run_ours(fv, first, count, nbr, (24, 24), 0.01, K, bin_size, 100, True, True, False)
            _assert_fwd_equal(ours, ref, tag=f"neighbor K={K} bin={bin_size}")


def test_empty_and_degenerate_inputs():
    from pytorch3d_amd import _C

    d = _dev()
    # a mesh with zero faces in the middle of the batch, a zero-area face, everything behind the camera
    gen = torch.Generator().manual_seed(0)
    fv = U.triangle_soup(30, gen)
    fv[3, 2] = fv[3, 1]  # zero area
    fv[10:20, :, 2] = -1.0  # behind
    first = torch.tensor([0, 10, 10, 20])
    count = torch.tensor([10, 0, 10, 10])
    nbr = torch.full((30,), -1, dtype=torch.int64)
    ref = orc.rasterize_meshes_naive(fv, first, count, nbr, (16, 16), 0.001, 3, True, True, False)
    for bin_size in (0, 8):
        ours = _run_ours(fv, first, count, nbr, (16, 16), 0.001, 3, bin_size, 50, True, True, False)
        _assert_fwd_equal(ours, ref, tag="degenerate")
    assert (ours[0][1] == -1).all() and (ours[0][2] == -1).all()
    # no faces at all / K == 0 / N == 0
    e = torch.zeros((0, 3, 3), device=d)
    z1 = torch.zeros(1, dtype=torch.int64, device=d)
    out = _C.rasterize_meshes(e, z1, z1, torch.zeros(0, dtype=torch.int64, device=d), (8, 8), 0.0, 2, 8, 10, False,
                              False, False)
    assert (out[0] == -1).all() and (out[1] == -1).all() and (out[2] == -1).all() and (out[3] == -1).all()
    out = _C.rasterize_meshes(fv.to(d), first.to(d), count.to(d), nbr.to(d), (8, 8), 0.0, 0, 0, 0, False, False, False)
    assert out[0].shape == (4, 8, 8, 0)


def test_coarse_bins_vs_oracle():
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(7)
    for (H, W), bin_size in [((32, 32), 8), ((64, 64), 16), ((24, 56), 8), ((50, 30), 5), ((128, 128), 8)]:
        for blur in (0.0, 0.01):
            F = 2500
            fv = U.triangle_soup(F, gen, size=0.3, behind_every=13)
            first = torch.tensor([0, 1100, 1100])
            count = torch.tensor([1100, 0, 1400])
            M = 400
            ref, ovf = orc.rasterize_meshes_coarse(fv, first, count, (H, W), blur, bin_size, M)
            ours = _C._rasterize_meshes_coarse(fv.to(d), first.to(d), count.to(d), (H, W), blur, bin_size, M).cpu()
            assert torch.equal(ours, ref), f"bins differ {(H, W)} {bin_size} {blur}"
    # overflow: first M ascending are kept
    ref, ovf = orc.rasterize_meshes_coarse(fv, first, count, (32, 32), 0.05, 16, 20)
    assert ovf
    ours = _C._rasterize_meshes_coarse(fv.to(d), first.to(d), count.to(d), (32, 32), 0.05, 16, 20).cpu()
    assert torch.equal(ours, ref)


@pytest.mark.parametrize("N,F,size,bin_size", [(70, 2100, (32, 32), 8),     # > 64 elements: the plan kernel, thread-per-row scan
                                               (65, 9000, (48, 40), 8),      # ragged, with empty elements
                                               (3, 150000, (64, 64), 16),    # 49+ chunks per element: the wave-per-row scan
                                               (64, 1900, (32, 32), 16),     # 64 elements: count pass publishes the plan
                                               (1, 150000, (80, 48), 16),    # one image, 147 chunks: the row scan with the offsets scan
                                               (1, 140000, (336, 336), 16)]) # as its tail (round 6); 15 rows in 4 workgroups / 441 rows in 111
def test_coarse_bins_of_the_launch_shapes_of_bin_build(N, F, size, bin_size):
    """binning.hip: bin_build picks its launches by batch size (plan kernel above 64 elements, otherwise the count pass's first
    workgroup publishes the chunk table), by rows / chunks (single-workgroup scan for small launches) and by chunks per element
    (thread-per-row scan fused with the block sums, or wave-per-row scan + block sums): bins vs the oracle for each shape,
    and the fused operator (internal tile bins, tile plan) vs our own naive kernel."""
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(N + F)
    fv = U.triangle_soup(F, gen, size=0.25 if F < 100000 else 0.05, behind_every=17)
    first, count = U.split_counts(F, N)
    if N == 65:  # two empty elements, one of them last
        count = count.clone()
        count[7] = 0
        count[64] = 0
    M = 300
    ref, _ = orc.rasterize_meshes_coarse(fv, first, count, size, 0.005, bin_size, M)
    ours = _C._rasterize_meshes_coarse(fv.to(d), first.to(d), count.to(d), size, 0.005, bin_size, M).cpu()
    assert torch.equal(ours, ref)
    nbr = torch.full((F,), -1, dtype=torch.int64)
    naive = _run_ours(fv, first, count, nbr, size, 0.005, 4, 0, 0, True, True, False)
    binned = _run_ours(fv, first, count, nbr, size, 0.005, 4, bin_size, 60000, True, True, False)  # (no bin overflows at this M)
    _assert_fwd_equal(binned, naive, tag=f"N={N} F={F}")


def test_fine_from_user_bins_with_holes():
    """_rasterize_meshes_fine accepts -1 sentinels anywhere in bin_faces (rasterize_meshes.cu:693-697)."""
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(11)
    F = 200
    fv = U.triangle_soup(F, gen)
    first, count = U.split_counts(F, 2)
    nbr = torch.full((F,), -1, dtype=torch.int64)
    bins, _ = orc.rasterize_meshes_coarse(fv, first, count, (32, 32), 0.01, 8, 100)
    # scatter the valid entries inside each row (order kept), so holes appear in the middle
    N, BH, BW, M = bins.shape
    holes = torch.full((N, BH, BW, 2 * M), -1, dtype=torch.int32)
    holes[..., ::2] = bins
    ref = orc.rasterize_meshes_naive(fv, first, count, nbr, (32, 32), 0.01, 4, True, False, False)
    ours = _C._rasterize_meshes_fine(fv.to(d), holes.to(d), nbr.to(d), (32, 32), 0.01, 8, 4, True, False, False)
    _assert_fwd_equal([o.cpu() for o in ours], ref, tag="fine-with-holes")


@pytest.mark.parametrize("persp,clip", [(False, False), (True, False), (False, True), (True, True)])
def test_backward_vs_oracle(persp, clip):
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(21)
    verts, faces = U.hetero_batch(3, seed=4, fmin=200, fmax=800)
    from pytorch3d_amd import PackedMeshes

    m = PackedMeshes(verts, faces)
    fv = m.verts_packed()[m.faces_packed()]
    first, count = m.mesh_to_faces_packed_first_idx(), m.num_faces_per_mesh()
    nbr = torch.full((fv.shape[0],), -1, dtype=torch.int64)
    size, K, blur = (48, 48), 4, 1e-3
    fwd = _run_ours(fv, first, count, nbr, size, blur, K, 8, 1000, persp, clip, False)
    ref_fwd = orc.rasterize_meshes_naive(fv, first, count, nbr, size, blur, K, persp, clip, False)
    _assert_fwd_equal(fwd, ref_fwd, tag="bwd-setup")
    gz = torch.randn(fwd[1].shape, generator=gen)
    gb = torch.randn(fwd[2].shape, generator=gen)
    gd = torch.randn(fwd[3].shape, generator=gen)
    ref = orc.rasterize_meshes_backward(fv, fwd[0], gz, gb, gd, persp, clip, cuda_semantics=True, acc64=True)
    ours = _C.rasterize_meshes_backward(fv.to(d), fwd[0].to(d), gz.to(d), gb.to(d), gd.to(d), persp, clip).cpu()
    # gate: the float64 restatement of the reference's formulas, per entry against the sum of the absolute per-sample terms
    # (tests/_util.py); the C oracle's float32 result is judged by the same gate and must agree where it passes
    U.assert_face_grads_vs_truth(f"backward persp={persp} clip={clip}", ours, fv, fwd[0], gz, gb, gd, persp, clip, rtol=2e-3,
                                 reference=ref)


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 8, 12, 16, 20, 32, 40])
def test_backward_every_kernel_vs_oracle(K):
    """Every backward kernel of the launcher (raster_mesh_bwd.hip: the sample-major rows kernel for K = 4, 8, 16 and -- round 4 --
    32, the pixel-major one for every other K) on the fragments of a dense soup, with and without the forward's row cover,
    against the float64 restatement of the reference's formulas; an image whose sides are no multiple of the 16-pixel areas."""
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(100 + K)
    fv = U.smooth_soup(600, gen, size=4.0)  # frame-sized triangles: rows of 40 slots fill up
    first, count = U.split_counts(600, 2)
    nbr = torch.full((600,), -1, dtype=torch.int64)
    size, blur = (40, 56), 2e-3
    (p2f, zbuf, bary, dists), cover = _C._rasterize_meshes_covered(fv.to(d), first.to(d), count.to(d), nbr.to(d), size, blur, K, 8, 1000,
                                                                    True, True, False)
    assert int((p2f[..., K - 1] >= 0).sum()) > 0, "the soup does not fill a row of this length"
    gz = torch.randn(zbuf.shape, generator=gen).to(d)
    gb = torch.randn(bary.shape, generator=gen).to(d)
    gd = torch.randn(dists.shape, generator=gen).to(d)
    plain = _C.rasterize_meshes_backward(fv.to(d), p2f.clone(), gz, gb, gd, True, True)
    covered = _C.rasterize_meshes_backward(fv.to(d), p2f, gz, gb, gd, True, True, _cover=cover)
    for tag, got in (("no cover", plain), ("row cover", covered)):
        U.assert_face_grads_vs_truth(f"backward K={K} {tag}", got.cpu(), fv, p2f.cpu(), gz.cpu(), gb.cpu(), gd.cpu(), True, True, rtol=2e-3)


@pytest.mark.parametrize("K", [4, 8])
def test_backward_with_per_face_reciprocals(K):
    """p3d_gather_face_verts_pre + p3d_rasterize_meshes_backward_verts_pre (round 6): the gather writes 1 / area and 1 / |edge|^2 per
    face (-1 for an edge of squared length <= 1e-8: geometry_utils.cuh:345), the perspective + clip backward reads them instead of
    forming them per sample.  A soup with needles (one degenerate edge each) and slivers: the records against torch, the gradient
    against the float64 restatement and against the backward that forms them per sample, with and without the cover list."""
    from pytorch3d_amd import _C, _lib

    d = _dev()
    lib = _lib.load()
    gen = torch.Generator().manual_seed(900 + K)
    F = 700
    fv = U.smooth_soup(F, gen, size=3.0)
    for j, i in enumerate(torch.randperm(F, generator=gen)[:60].tolist()):
        a, b = (0, 1) if j % 3 == 0 else ((1, 2) if j % 3 == 1 else (2, 0))
        step = (torch.rand(2, generator=gen) - 0.5) * (3e-4 if j % 2 else 2e-5)  # both sides of the 1e-4 threshold
        fv[i, b, :2] = fv[i, a, :2] + step
    verts = fv.reshape(-1, 3).contiguous().to(d)  # every face owns its vertices: grad_verts IS grad_face_verts
    faces = torch.arange(3 * F, dtype=torch.int64).reshape(F, 3).to(d)
    faces[5] = faces[5] - 3 * F  # negative ids wrap once, as torch indexing
    first, count = U.split_counts(F, 2)
    nbr = torch.full((F,), -1, dtype=torch.int64)
    face_verts = torch.empty((F, 3, 3), device=d)
    pre = torch.empty((F, 4), device=d)
    _lib.check(lib.p3d_gather_face_verts_pre(_C._ptr(verts), _C._ptr(faces), 3 * F, F, _C._ptr(face_verts), _C._ptr(pre),
                                             _C._stream(d)), "gather_face_verts_pre")
    assert torch.equal(face_verts.cpu(), fv)
    v0, v1, v2 = fv[:, 0, :2], fv[:, 1, :2], fv[:, 2, :2]
    cross = lambda a, b: a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]  # float32, product by product: geometry_utils.cuh:63
    area = (cross(v2 - v0, v1 - v0).double() + 1e-8).float().double()  # edge function of (v2; v0, v1) + kEpsilon, :108
    want = [1.0 / area]
    for a, b in ((v0, v1), (v0, v2), (v1, v2)):
        l2 = ((b - a) ** 2).sum(1)
        want.append(torch.where(l2 <= 1e-8, torch.full_like(l2, -1.0), 1.0 / l2).double())
    got = pre.cpu().double()
    assert int((got[:, 1:] == -1).sum()) > 20, "no degenerate edge in the soup"
    for c in range(4):
        sel = (want[c] == -1) | (got[:, c] == -1)
        assert torch.equal(got[sel, c], want[c][sel]), f"record column {c}: the degenerate edges differ"
        rel = ((got[~sel, c] - want[c][~sel]).abs() / want[c][~sel].abs()).max()
        assert float(rel) < 2e-6, (c, float(rel))

    size, blur = (40, 56), 2e-3
    (p2f, zbuf, bary, dists), cover = _C._rasterize_meshes_covered(face_verts, first.to(d), count.to(d), nbr.to(d), size, blur, K, 8, 1000,
                                                                    True, True, False)
    assert _C.cover_has_list(cover, 2, *size)
    gz = torch.randn(zbuf.shape, generator=gen).to(d)
    gb = torch.randn(bary.shape, generator=gen).to(d)
    gd = torch.randn(dists.shape, generator=gen).to(d)

    def backward(with_pre, with_cover):
        out = torch.full((3 * F, 3), float("nan"), device=d)
        rc = lib.p3d_rasterize_meshes_backward_verts_pre(
            _C._ptr(face_verts), _C._ptr(pre) if with_pre else None, _C._ptr(faces), _C._ptr(p2f), _C._ptr(gz), _C._ptr(gb), _C._ptr(gd),
            _C.cover_ptr(cover, 2, *size) if with_cover else None, F, 3 * F, 2, size[0], size[1], K, 1, 1, _C._ptr(out), _C._stream(d))
        _lib.check(rc, "backward_verts_pre")
        return out.cpu().reshape(F, 3, 3)  # (face 5's wrapped ids address its own rows)

    plain = backward(False, True)
    for with_cover in (True, False):
        got = backward(True, with_cover)
        U.assert_face_grads_vs_truth(f"backward with face records K={K} cover={with_cover}", got, fv, p2f.cpu(), gz.cpu(), gb.cpu(),
                                     gd.cpu(), True, True, rtol=2e-3, reference=plain)
    # not perspective + clip: the records are ignored, not misread
    out = torch.empty((3 * F, 3), device=d)
    _lib.check(lib.p3d_rasterize_meshes_backward_verts_pre(
        _C._ptr(face_verts), _C._ptr(pre), _C._ptr(faces), _C._ptr(p2f), _C._ptr(gz), _C._ptr(gb), _C._ptr(gd), None, F, 3 * F, 2,
        size[0], size[1], K, 0, 0, _C._ptr(out), _C._stream(d)), "backward_verts_pre")
    want_ff = _C.rasterize_meshes_backward(face_verts, p2f.clone(), gz, gb, gd, False, False)
    U.assert_face_grads_vs_truth(f"backward with face records K={K}, flat", out.cpu().reshape(F, 3, 3), fv, p2f.cpu(), gz.cpu(), gb.cpu(),
                                 gd.cpu(), False, False, rtol=2e-3, reference=want_ff.cpu())


@pytest.mark.parametrize("K", [4, 8])
def test_reference_signature_backward_with_and_without_face_records(K):
    """`_C.rasterize_meshes_backward` as the reference calls it (face_verts in, grad_face_verts out): with _C.FACE_PRE the call first
    writes the per-face reciprocals (p3d_rasterize_meshes_backward_pre) -- same gradients as the per-sample form within the gate, with
    the forward's cover (list), a cloned cover (no list: workspace) and none."""
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(300 + K)
    fv = U.smooth_soup(500, gen, size=3.0)
    first, count = U.split_counts(500, 2)
    nbr = torch.full((500,), -1, dtype=torch.int64)
    size = (48, 40)
    (p2f, zbuf, bary, dists), cover = _C._rasterize_meshes_covered(fv.to(d), first.to(d), count.to(d), nbr.to(d), size, 2e-3, K, 8, 1000,
                                                                    True, True, False)
    gz, gb, gd = (torch.randn(t.shape, generator=gen).to(d) for t in (zbuf, bary, dists))
    saved = _C.FACE_PRE
    try:
        res = {}
        for pre in (True, False):
            _C.FACE_PRE = pre
            res[pre] = [_C.rasterize_meshes_backward(fv.to(d), p2f, gz, gb, gd, True, True, _cover=c).cpu()
                        for c in (cover, cover.clone(), None)]
    finally:
        _C.FACE_PRE = saved
    for got in res[True]:
        U.assert_face_grads_vs_truth(f"reference-signature backward with face records K={K}", got, fv, p2f.cpu(), gz.cpu(), gb.cpu(), gd.cpu(),
                                     True, True, rtol=2e-3, reference=res[False][0])


def test_autograd_mirror_and_reference_cpu_build():
    """The L2 mirror end to end (verts -> loss -> grad), against the reference's own CPU kernels when
    oracle/_ref is present (idx exact, floats 1e-5, grads rtol 5e-3 as tests/test_rasterize_meshes.py:317-319)."""
    import pytorch3d_amd as p3d

    d = _dev()
    verts, faces = U.hetero_batch(2, seed=9, fmin=300, fmax=1200)
    vg = [v.to(d).requires_grad_(True) for v in verts]
    meshes = p3d.PackedMeshes(vg, [f.to(d) for f in faces])
    out = p3d.rasterize_meshes(meshes, image_size=64, blur_radius=1e-4, faces_per_pixel=4, perspective_correct=True,
                               clip_barycentric_coords=False)
    gen = torch.Generator().manual_seed(231)
    g = [torch.randn(o.shape, generator=gen).to(d) for o in out[1:]]
    torch.autograd.backward(list(out[1:]), g)
    ref = orc.ref_module()
    if ref is None:
        pytest.skip("oracle/_ref not built on this box")
    mc = p3d.PackedMeshes(verts, faces)
    fv = mc.verts_packed()[mc.faces_packed()].clone().requires_grad_(False)
    nbr = torch.full((fv.shape[0],), -1, dtype=torch.int64)
    r = ref._rasterize_meshes_naive(fv, mc.mesh_to_faces_packed_first_idx(), mc.num_faces_per_mesh(), nbr, (64, 64),
                                    1e-4, 4, True, False, False)
    # The reference's CPU kernels multiply the perspective-correction numerators in a different order
    # than its CUDA kernels (geometry_utils.h:200 vs geometry_utils.cuh:179), so depths can differ by
    # an ulp and two faces that tie at a shared edge may swap places.  Indices must agree except at
    # such ties; zbuf agrees everywhere; bary / dists where the index agrees.
    ours = [o.detach().cpu() for o in out]
    same = ours[0] == r[0]
    set_same = (ours[0].sort(-1).values == r[0].sort(-1).values).all(-1)
    n_set, n_idx = int((~set_same).sum()), int((~same).sum())
    print(f"[mirror vs reference CPU build] pixels whose face SET differs {n_set} / {set_same.numel()}, slots whose index differs "
          f"{n_idx} / {same.numel()} (tie swaps at shared edges, see above)")
    # observed on MI355X (round 3): 4 of 8192 pixels with a different face set, 138 of 32768 slots with a different index
    # (0.4 %: this scene is mostly shared edges at 64x64); the gates are twice that
    assert n_set <= 8, n_set
    assert n_idx <= 276, n_idx
    assert torch.allclose(ours[1], r[1], atol=1e-5, rtol=0)
    assert torch.allclose(ours[2][same], r[2][same], atol=1e-5, rtol=0)
    assert torch.allclose(ours[3][same], r[3][same], atol=1e-5, rtol=0)
    # exactness is against the CUDA-order oracle
    o = orc.rasterize_meshes_naive(fv, mc.mesh_to_faces_packed_first_idx(), mc.num_faces_per_mesh(), nbr, (64, 64),
                                   1e-4, 4, True, False, False)
    assert all(torch.equal(a, b) for a, b in zip(ours, o))
    ours_fwd_idx = ours[0]
    rg = ref.rasterize_meshes_backward(fv, ours[0], g[0].cpu(), g[1].cpu(), g[2].cpu(), True, False)
    # scatter reference face grads to verts the way autograd does
    gv = torch.zeros_like(mc.verts_packed())
    gv.index_add_(0, mc.faces_packed().reshape(-1), rg.reshape(-1, 3))
    ours = torch.cat([v.grad.cpu() for v in vg], 0)
    U.assert_face_grads_vs_truth("autograd mirror, gradient to the vertices", ours, fv, ours_fwd_idx, g[0].cpu(), g[1].cpu(), g[2].cpu(),
                                 True, False, reference=gv, faces=mc.faces_packed(), num_verts=gv.shape[0])


def test_large_image_property_checks():
    """At the benchmark's resolution (512^2, K=8) the oracle is too slow; check properties instead:
    naive == binned, K sorted by z, padding is exactly -1, bary sums to 1 where clipped."""
    verts, faces = U.hetero_batch(2, seed=1, fmin=2000, fmax=6000)
    from pytorch3d_amd import PackedMeshes

    m = PackedMeshes(verts, faces)
    fv = m.verts_packed()[m.faces_packed()]
    first, count = m.mesh_to_faces_packed_first_idx(), m.num_faces_per_mesh()
    nbr = torch.full((fv.shape[0],), -1, dtype=torch.int64)
    blur = 9.2e-4
    a = _run_ours(fv, first, count, nbr, (512, 512), blur, 8, 32, 10000, True, True, False)
    b = _run_ours(fv, first, count, nbr, (512, 512), blur, 8, 0, 0, True, True, False)
    _assert_fwd_equal(a, b, tag="naive==binned@512")
    p2f, zbuf, bary, dists = a
    valid = p2f >= 0
    assert (zbuf[~valid] == -1).all() and (dists[~valid] == -1).all() and (bary[~valid] == -1).all()
    # valid entries form a prefix along K and are sorted by z
    assert (valid[..., 1:] <= valid[..., :-1]).all()
    z = torch.where(valid, zbuf, torch.full_like(zbuf, float("inf")))
    assert (z[..., 1:] >= z[..., :-1]).all()
    s = bary.sum(-1)[valid]
    assert torch.allclose(s, torch.ones_like(s), atol=1e-4)
    # faces belong to the right mesh
    for n in range(2):
        f = p2f[n][valid[n]]
        assert (f >= first[n]).all() and (f < first[n] + count[n]).all()
    assert valid.float().mean() > 0.02


def test_gather_scatter_face_verts_vs_torch_indexing():
    """p3d_gather_face_verts / p3d_scatter_face_grads vs `verts_packed[faces_packed]` and its autograd
    (pytorch3d/renderer/mesh/rasterize_meshes.py:146): forward bit-exact, backward to float-sum reordering."""
    import importlib

    rm = importlib.import_module("pytorch3d_amd.rasterize_meshes")
    d = _dev()
    gen = torch.Generator().manual_seed(3)
    V, F = 5000, 12000
    verts = torch.randn(V, 3, generator=gen).to(d)
    faces = torch.randint(0, V, (F, 3), generator=gen).to(d)
    v1 = verts.clone().requires_grad_(True)
    v2 = verts.clone().requires_grad_(True)
    a = rm.gather_face_verts(v1, faces)
    b = v2[faces]
    assert torch.equal(a, b)
    g = torch.randn(F, 3, 3, generator=gen).to(d)
    a.backward(g)
    b.backward(g)
    assert torch.allclose(v1.grad, v2.grad, rtol=1e-5, atol=1e-5)
    # unused vertices get exact zeros; an unused output gradient is handled (materialize_grads off)
    assert torch.equal(v1.grad == 0, v2.grad == 0)
    e = rm.gather_face_verts(verts[:0].clone().requires_grad_(True), faces[:0])
    assert e.shape == (0, 3, 3)


def test_backward_with_unused_outputs():
    """Only zbuf feeds the loss: grad_bary / grad_dists arrive as None (set_materialize_grads(False))."""
    import pytorch3d_amd as p3d

    d = _dev()
    verts, faces = U.hetero_batch(1, seed=2, fmin=200, fmax=400)
    vg = [v.to(d).requires_grad_(True) for v in verts]
    out = p3d.rasterize_meshes(p3d.PackedMeshes(vg, [f.to(d) for f in faces]), image_size=32, blur_radius=1e-3,
                               faces_per_pixel=4, perspective_correct=True, clip_barycentric_coords=True)
    out[1].sum().backward()
    gz = torch.ones_like(out[1]).cpu()
    fv = verts[0][faces[0]]
    ref = orc.rasterize_meshes_backward(fv, out[0].cpu(), gz, torch.zeros(out[2].shape), torch.zeros(out[3].shape), True,
                                        True)
    gv = torch.zeros_like(verts[0])
    gv.index_add_(0, faces[0].reshape(-1), ref.reshape(-1, 3))
    U.assert_face_grads_vs_truth("only zbuf feeds the loss", vg[0].grad.cpu(), fv, out[0].cpu(), gz, torch.zeros(out[2].shape),
                                 torch.zeros(out[3].shape), True, True, reference=gv, faces=faces[0], num_verts=gv.shape[0])


def test_rasterize_meshes_is_hip_graph_capturable():
    """The C ABI launches are asynchronous on the caller's stream, allocate nothing and never synchronise: a
    torch.cuda.graph capture of the operator replays bit-identical results (profiles/graph_c2.py)."""
    from pytorch3d_amd import _C

    d = torch.device("cuda:0")
    v, f = U.ico_sphere(3)
    fv = U.to_ndc(v)[f].to(d).contiguous()
    F = fv.shape[0]
    first = torch.zeros(1, dtype=torch.int64, device=d)
    cnt = torch.tensor([F], dtype=torch.int64, device=d)
    nbr = torch.full((F,), -1, dtype=torch.int64, device=d)
    args = (fv, first, cnt, nbr, (96, 128), 1e-4, 4, 16, 2000, True, True, False)
    ref = _C.rasterize_meshes(*args)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        _C.rasterize_meshes(*args)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = _C.rasterize_meshes(*args)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    for a, b in zip(out, ref):
        assert torch.equal(a, b)


@pytest.mark.parametrize("size,K", [((1000, 1400), 4), ((2048, 640), 8), ((1300, 1300), 5)])
def test_images_larger_than_the_bin_grid_property_checks(size, K):
    """Above 512 pixels the internal bins grow beyond one 16x16 tile (at most 32 bins per side): naive == binned, sorted
    K-prefixes, -1 padding, also for non-square sizes that are no multiple of the tile and for a K without vector rows;
    the backward of both agrees."""
    verts, faces = U.hetero_batch(2, seed=7, fmin=1500, fmax=4000)
    from pytorch3d_amd import PackedMeshes, _C

    m = PackedMeshes(verts, faces)
    fv = m.verts_packed()[m.faces_packed()]
    first, count = m.mesh_to_faces_packed_first_idx(), m.num_faces_per_mesh()
    nbr = torch.full((fv.shape[0],), -1, dtype=torch.int64)
    blur = 3e-4
    from pytorch3d_amd.rasterize_meshes import default_bin_size

    a = _run_ours(fv, first, count, nbr, size, blur, K, default_bin_size(max(size)), 20000, True, True, False)
    b = _run_ours(fv, first, count, nbr, size, blur, K, 0, 0, True, True, False)
    _assert_fwd_equal(a, b, tag=f"naive==binned@{size}")
    p2f, zbuf, bary, dists = a
    valid = p2f >= 0
    assert (zbuf[~valid] == -1).all() and (dists[~valid] == -1).all() and (bary[~valid] == -1).all()
    assert (valid[..., 1:] <= valid[..., :-1]).all()
    z = torch.where(valid, zbuf, torch.full_like(zbuf, float("inf")))
    assert (z[..., 1:] >= z[..., :-1]).all()
    assert valid.float().mean() > 0.01
    # backward: linear in the upstream gradients (size-independent property), finite, and only on faces that were hit
    d = _dev()
    gen = torch.Generator().manual_seed(K)
    gz, gd = (torch.randn(p2f.shape, generator=gen).to(d) for _ in range(2))
    gb = torch.randn(p2f.shape + (3,), generator=gen).to(d)
    fvd, p2fd = fv.to(d), p2f.to(d)
    g1 = _C.rasterize_meshes_backward(fvd, p2fd, gz, gb, gd, True, True)
    g2 = _C.rasterize_meshes_backward(fvd, p2fd, 2 * gz, 2 * gb, 2 * gd, True, True)
    assert torch.isfinite(g1).all()
    assert torch.allclose(g2, 2 * g1, rtol=1e-3, atol=1e-3 * g1.abs().max().item())
    hit = torch.zeros(fv.shape[0], dtype=torch.bool)
    hit[p2f[valid]] = True
    assert (g1.cpu()[~hit] == 0).all()


@pytest.mark.parametrize("size", [(1000, 1000), (720, 1280), (528, 528), (1000, 1400)])
@pytest.mark.parametrize("K", [4, 8])
def test_large_images_background_is_written(size, K, monkeypatch):
    """above 512 pixels a side an internal bin holds several 16x16 tiles; when the image leaves a
    partial last bin row / column (1000 -> 32-pixel bins, last one 8 pixels; 720x1280 -> 64-pixel bins, last row 16) and
    the mesh reaches that border, the piggyback fill of round 3 left tiles of background bins unwritten (the outputs are
    torch.empty).  The outputs are pre-filled with a sentinel here; the meshes overflow the frame on every side; binned
    must equal naive and no sentinel may survive."""
    from pytorch3d_amd import PackedMeshes, _C

    verts, faces = U.hetero_batch(2, seed=11, fmin=600, fmax=2500, torus_div=0.7)
    m = PackedMeshes(verts, faces)
    fv = m.verts_packed()[m.faces_packed()]
    first, count = m.mesh_to_faces_packed_first_idx(), m.num_faces_per_mesh()
    nbr = torch.full((fv.shape[0],), -1, dtype=torch.int64)
    real = _C._mesh_outputs

    def sentinel_outputs(N, H, W, K_, device):
        out = real(N, H, W, K_, device)
        out[0].fill_(7777777)
        for o in out[1:]:
            o.fill_(123.0)
        return out

    monkeypatch.setattr(_C, "_mesh_outputs", sentinel_outputs)
    from pytorch3d_amd.rasterize_meshes import default_bin_size

    a = _run_ours(fv, first, count, nbr, size, 3e-4, K, default_bin_size(max(size)), 20000, True, True, False)
    b = _run_ours(fv, first, count, nbr, size, 3e-4, K, 0, 0, True, True, False)
    assert int((a[0] == 7777777).sum()) == 0, "pix_to_face entries never written by the binned kernel"
    for o in a[1:]:
        assert int((o == 123.0).sum()) == 0, "float outputs never written by the binned kernel"
    _assert_fwd_equal(a, b, tag=f"naive==binned@{size}")
    # the meshes reach all four borders of the output (what makes the partial last bins active)
    hit = a[0][..., 0] >= 0
    assert hit[:, 0, :].any() and hit[:, -1, :].any() and hit[:, :, 0].any() and hit[:, :, -1].any()


def test_operators_are_reentrant_across_threads_and_streams():
    """The launchers keep no global mutable state and launch on the caller's current stream (nn.DataParallel calls them
    from several Python threads, tests/test_render_multigpu.py:171 in the reference): four threads, each on its own
    stream, rasterizing + differentiating different meshes concurrently, reproduce the single-threaded results."""
    import threading

    from pytorch3d_amd import _C

    d = _dev()
    jobs = []
    for seed in range(4):
        verts, faces = U.hetero_batch(1, seed=20 + seed, fmin=800, fmax=3000)
        fv = verts[0][faces[0]].to(d).contiguous()
        F = fv.shape[0]
        jobs.append((fv, torch.zeros(1, dtype=torch.int64, device=d), torch.tensor([F], dtype=torch.int64, device=d),
                     torch.full((F,), -1, dtype=torch.int64, device=d)))
    gen = torch.Generator().manual_seed(0)
    size, K = (192, 160), 4
    gz = torch.randn(1, *size, K, generator=gen).to(d)
    gb = torch.randn(1, *size, K, 3, generator=gen).to(d)

    def run(job):
        fv, first, count, nbr = job
        out = _C.rasterize_meshes(fv, first, count, nbr, size, 2e-4, K, 16, 5000, True, True, False)
        g = _C.rasterize_meshes_backward(fv, out[0], gz, gb, gz, True, True)
        return out, g

    ref = [run(j) for j in jobs]
    torch.cuda.synchronize()
    got = [None] * 4
    err = []

    def worker(i):
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.default_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    got[i] = run(jobs[i])
            s.synchronize()
        except Exception as e:  # surfaced below
            err.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not err, err
    for (o_ref, g_ref), (o, g) in zip(ref, got):
        assert all(torch.equal(a, b) for a, b in zip(o_ref, o))
        assert torch.allclose(g, g_ref, rtol=1e-4, atol=1e-5 * g_ref.abs().max().item())


def test_gather_face_verts_index_and_shape_rules():
    """`verts_packed[faces_packed]` replacement: negative ids wrap like torch indexing, ids out of range never touch memory
    outside `verts` (NaN forward, dropped in the backward), and anything that is not (V,3) x (F,3) takes torch indexing."""
    from pytorch3d_amd.rasterize_meshes import gather_face_verts

    d = _dev()
    gen = torch.Generator().manual_seed(4)
    V, F = 50, 80
    verts = torch.randn(V, 3, generator=gen).to(d).requires_grad_(True)
    faces = torch.randint(0, V, (F, 3), generator=gen).to(d)
    neg = faces.clone()
    neg[::3] -= V  # the same vertices, addressed from the end
    a = gather_face_verts(verts, neg)
    assert torch.equal(a, verts[faces])
    g = torch.randn(F, 3, 3, generator=gen).to(d)
    (ga,) = torch.autograd.grad(a, verts, g)
    (gr,) = torch.autograd.grad(verts[faces], verts, g)
    assert torch.allclose(ga, gr, atol=1e-5)
    bad = faces.clone()
    bad[5, 1] = V + 7
    bad[9, 2] = -V - 3
    b = gather_face_verts(verts, bad)
    assert torch.isnan(b[5, 1]).all() and torch.isnan(b[9, 2]).all()
    ok = torch.ones(F, 3, dtype=torch.bool, device=d)
    ok[5, 1] = ok[9, 2] = False
    assert torch.equal(b[ok], verts[faces][ok])
    (gb,) = torch.autograd.grad(b, verts, g)
    assert torch.isfinite(gb).all()
    # (V, C != 3) attributes and non-triangles: plain torch semantics
    attrs = torch.randn(V, 5, generator=gen).to(d)
    assert torch.equal(gather_face_verts(attrs, faces), attrs[faces])
    quads = torch.randint(0, V, (F, 4), generator=gen).to(d)
    assert torch.equal(gather_face_verts(verts, quads), verts[quads])
