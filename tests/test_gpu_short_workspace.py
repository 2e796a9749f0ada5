"""Short workspaces of rasterize_meshes (include/p3d_amd.h "Short workspaces", csrc/binning.h, pytorch3d_amd/_C.py).

The coarse stage's lists are sized from what the same call shape needed before instead of for the worst case; whether they fit
is decided on the device (no host sync), and when they do not the naive kernel writes the output instead of the binned one.

  * a call whose lists do NOT fit (first guess of one entry) returns the bits of the call with the worst-case workspace --
    pix_to_face, zbuf, bary, dists and the row cover -- on every kernel class (payload queues, queues without payload,
    the private-memory queue, split mode for a single image), with and without blur, clipping flags, culling;
  * the call after it has learned the size: its workspace is a small fraction of the worst case, its lists fit (the binned
    kernel ran: the needed-entries word equals the list total of the worst-case call) and the bits are again the same;
  * the same for rasterize_points (every queue class of the point kernels);
  * a workspace smaller than the fixed arrays is refused with the C ABI's workspace error;
  * 'auto' leaves small batches on the worst-case workspace.
"""
import ctypes

import pytest
import torch

import _util as U

pytestmark = pytest.mark.gpu


def _batch(n, seed, fmin=300, fmax=1500):
    import pytorch3d_amd as p3d

    d = torch.device("cuda:0")
    verts, faces = U.hetero_batch(n, seed=seed, fmin=fmin, fmax=fmax)
    m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
    fv = m.verts_packed()[m.faces_packed()].contiguous()
    return fv, m.mesh_to_faces_packed_first_idx(), m.num_faces_per_mesh()


@pytest.fixture
def short_mode():
    from pytorch3d_amd import _C

    saved = (_C.SHORT_WORKSPACE, _C.SHORT_WORKSPACE_FIRST_GUESS)
    _C._NEEDS.clear()
    yield _C
    _C.SHORT_WORKSPACE, _C.SHORT_WORKSPACE_FIRST_GUESS = saved
    _C._NEEDS.clear()


def _same(a, b):
    return torch.equal(a[0], b[0]) and all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip(a[1:], b[1:]))


@pytest.mark.parametrize("K", [1, 3, 4, 8, 12, 40, 60])
@pytest.mark.parametrize("n,size,bin_size", [(1, (128, 128), 32), (6, (200, 200), 16), (3, (100, 77), 64)])
def test_overflowing_lists_fall_back_on_the_device_with_the_same_bits(short_mode, K, n, size, bin_size):
    _C = short_mode
    fv, first, cnt = _batch(n, seed=K + n)
    F = fv.shape[0]
    nbr = torch.full((F,), -1, dtype=torch.int64, device=fv.device)
    for blur, persp, clip, cull in ((0.0, False, False, False), (2e-3, True, True, False), (1e-3, True, False, True)):
        args = (fv, first, cnt, nbr, size, blur, K, bin_size, 4000, persp, clip, cull)
        _C.SHORT_WORKSPACE = "never"
        want, want_cover = _C._rasterize_meshes_covered(*args)
        worst = _C.WORKSPACE_STATS["last_bytes"]
        assert _C.WORKSPACE_STATS["last_entries"] is None

        _C.SHORT_WORKSPACE, _C.SHORT_WORKSPACE_FIRST_GUESS = "always", 1
        _C._NEEDS.clear()
        got, got_cover = _C._rasterize_meshes_covered(*args)  # lists do not fit: the naive kernel writes
        assert _C.WORKSPACE_STATS["last_entries"] == 1 and _C.WORKSPACE_STATS["last_bytes"] < worst
        assert _same(got, want) and torch.equal(got_cover, want_cover)
        torch.cuda.synchronize()

        got, got_cover = _C._rasterize_meshes_covered(*args)  # learned: lists fit, the binned kernel writes
        need = next(iter(_C._NEEDS.values())).entries
        assert need is not None and need > 0
        assert _C.WORKSPACE_STATS["last_entries"] == need + need // 4 + 4096
        assert _same(got, want) and torch.equal(got_cover, want_cover)
        # the needed-entries word is the total of the lists: the same through the test-visible coarse operator when the
        # caller's bins are the internal ones (bin_size 16 at these sizes)
        if bin_size == 16:
            bins = _C._rasterize_meshes_coarse(fv, first, cnt, size, blur, bin_size, 4000)
            assert need == int((bins >= 0).sum())


def test_short_workspace_is_a_fraction_of_the_worst_case_and_auto_keeps_small_batches_whole(short_mode):
    _C = short_mode
    fv, first, cnt = _batch(16, seed=5, fmin=2000, fmax=6000)
    F = fv.shape[0]
    nbr = torch.full((F,), -1, dtype=torch.int64, device=fv.device)
    args = (fv, first, cnt, nbr, (256, 256), 1e-4, 8, 32, max(10000, F // 5), True, True, False)
    _C.SHORT_WORKSPACE = "auto"
    want = _C.rasterize_meshes(*args)
    worst = _C.WORKSPACE_STATS["last_bytes"]
    assert _C.WORKSPACE_STATS["last_entries"] is None and worst < _C.SHORT_WORKSPACE_ABOVE  # small batch: untouched
    _C.SHORT_WORKSPACE = "always"
    calls = _C.WORKSPACE_STATS["short_calls"]
    for i in range(12):  # first guess (32 per face), then learned sizes; reports every call at first, then every 8th
        got = _C.rasterize_meshes(*args)
        assert _same(got, want)
        if i == 0:
            torch.cuda.synchronize()
    assert _C.WORKSPACE_STATS["short_calls"] == calls + 12
    assert _C.WORKSPACE_STATS["last_bytes"] * 8 < worst, (_C.WORKSPACE_STATS, worst)
    # 'auto' turns to short workspaces above the threshold
    _C.SHORT_WORKSPACE, above = "auto", _C.SHORT_WORKSPACE_ABOVE
    try:
        _C.SHORT_WORKSPACE_ABOVE = worst - 1
        got = _C.rasterize_meshes(*args)
        assert _C.WORKSPACE_STATS["last_entries"] is not None and _same(got, want)
    finally:
        _C.SHORT_WORKSPACE_ABOVE = above


@pytest.mark.parametrize("K", [1, 5, 8, 10, 40, 70, 120])
def test_points_overflowing_lists_fall_back_on_the_device_with_the_same_bits(short_mode, K):
    """rasterize_points with a short workspace: first call with one list entry of room (the naive kernel writes), second
    call with the learned size (the binned kernel writes): idx, zbuf and dists equal the worst-case call's, bit for bit."""
    _C = short_mode
    d = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(K)
    sizes = [3000, 0, 5000]
    P = sum(sizes)
    pts = torch.rand((P, 3), generator=gen) * torch.tensor([2.0, 2.0, 2.0]) - torch.tensor([1.0, 1.0, -0.2])
    pts[::97, 2] = -0.5  # some behind the camera
    rad = torch.rand((P,), generator=gen) * 0.05 + 0.01
    first = torch.tensor([0, sizes[0], sizes[0]], dtype=torch.int64)
    count = torch.tensor(sizes, dtype=torch.int64)
    args = (pts.to(d), first.to(d), count.to(d), (96, 80), rad.to(d), K, 16, 3000)
    _C.SHORT_WORKSPACE = "never"
    want = _C.rasterize_points(*args)
    worst = _C.WORKSPACE_STATS["last_bytes"]
    _C.SHORT_WORKSPACE, _C.SHORT_WORKSPACE_FIRST_GUESS = "always", 1
    got = _C.rasterize_points(*args)
    assert _C.WORKSPACE_STATS["last_entries"] == 1 and _C.WORKSPACE_STATS["last_bytes"] < worst
    assert _same(got, want)
    torch.cuda.synchronize()
    got = _C.rasterize_points(*args)
    need = next(iter(_C._NEEDS.values())).entries
    assert need is not None and need > 0 and _C.WORKSPACE_STATS["last_entries"] == need + need // 4 + 4096
    assert _same(got, want)
    bins = _C._rasterize_points_coarse(args[0], args[1], args[2], (96, 80), args[4], 16, 3000)
    assert need == int((bins >= 0).sum())  # 16-pixel bins are the internal ones here: the word is the lists' total


def test_a_workspace_below_the_fixed_arrays_is_refused():
    from pytorch3d_amd import _C, _lib

    lib = _lib.load()
    fv, first, cnt = _batch(2, seed=1)
    F, N, H = fv.shape[0], 2, 64
    nbr = torch.full((F,), -1, dtype=torch.int64, device=fv.device)
    least = lib.p3d_rasterize_meshes_short_workspace_bytes(F, N, H, H, 16, 1000, 0)
    out = _C._mesh_outputs(N, H, H, 4, fv.device)
    for nbytes, ok in ((least, True), (least - 512, False)):
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=fv.device)
        rc = lib.p3d_rasterize_meshes(_C._ptr(fv), _C._ptr(first), _C._ptr(cnt), _C._ptr(nbr), F, N, H, H, 0.0, 4, 16, 1000, 0, 0, 0,
                                      _C._ptr(out[0]), _C._ptr(out[1]), _C._ptr(out[2]), _C._ptr(out[3]), _C._ptr(ws), nbytes,
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        if ok:
            _lib.check(rc, "rasterize_meshes")
            want = _C.rasterize_meshes(fv, first, cnt, nbr, (H, H), 0.0, 4, 16, 1000, False, False, False)
            assert _same(out, want)  # one list entry of room: the naive kernel wrote
        else:
            with pytest.raises(RuntimeError, match="workspace"):
                _lib.check(rc, "rasterize_meshes")
