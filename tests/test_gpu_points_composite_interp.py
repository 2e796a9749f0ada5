"""GPU parity: point rasterization, the three compositors and interpolate_face_attributes vs the
oracle (and the reference CPU build when present), through the C ABI."""
import pytest
import torch

import _util as U
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _cloud(P, gen, zlo=-0.2, zhi=2.0):
    return torch.cat([torch.rand(P, 2, generator=gen) * 2.4 - 1.2, torch.rand(P, 1, generator=gen) * (zhi - zlo) + zlo],
                     1)


@pytest.mark.parametrize("size", [(32, 32), (20, 48), (45, 31)])
@pytest.mark.parametrize("K", [1, 3, 8, 10, 16, 20, 32, 40, 50, 64, 100, 120])
def test_points_naive_and_binned_vs_oracle(size, K):
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(K * 100 + size[0])
    P = 800
    pts = _cloud(P, gen)
    first = torch.tensor([0, 300, 300])
    count = torch.tensor([300, 0, 500])
    radius = torch.rand(P, generator=gen) * 0.12 + 0.02
    ref = orc.rasterize_points_naive(pts, first, count, size, radius, K)
    for bin_size in (0, 8, 16):
        ours = _C.rasterize_points(pts.to(d), first.to(d), count.to(d), size, radius.to(d), K, bin_size, 500)
        ours = [o.cpu() for o in ours]
        assert torch.equal(ours[0], ref[0]), f"idx differs bin={bin_size}: {(ours[0] != ref[0]).sum().item()}"
        assert torch.equal(ours[1], ref[1]) and torch.equal(ours[2], ref[2])
    ref_mod = orc.ref_module()
    if ref_mod is not None:
        r = ref_mod._rasterize_points_naive(pts, first, count, size, radius, K)
        assert all(torch.equal(a, b) for a, b in zip(ours, r))


@pytest.mark.parametrize("K", [1, 2, 4, 5, 8, 10, 12, 16, 17, 24, 25, 28, 29, 32, 33, 40, 50, 64, 65, 80, 99, 100, 101, 150])
def test_points_queues_overflow_vs_oracle(K):
    """A cloud dense enough that EVERY queue overflows (~200 splats over each pixel; the sparse cloud above never fills a
    queue beyond 8 entries): all capacities of the launcher, the pair queue of 100 entries with 65..99 live ones
    (round 4: TopKReg<100, 0> spilled 465 VGPRs and is gone), the private-memory queue beyond 100."""
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(1000 + K)
    P = 1500
    pts = _cloud(P, gen)
    first = torch.tensor([0, 700])
    count = torch.tensor([700, 800])
    radius = torch.rand(P, generator=gen) * 0.3 + 0.45
    size = (20, 37)
    ref = orc.rasterize_points_naive(pts, first, count, size, radius, K)
    assert int((ref[0][..., K - 1] >= 0).sum()) > 0, "the cloud does not fill a queue of this length"
    for bin_size in (0, 8):
        ours = _C.rasterize_points(pts.to(d), first.to(d), count.to(d), size, radius.to(d), K, bin_size, 1000)
        ours = [o.cpu() for o in ours]
        assert torch.equal(ours[0], ref[0]), f"idx differs K={K} bin={bin_size}: {(ours[0] != ref[0]).sum().item()}"
        assert torch.equal(ours[1], ref[1]) and torch.equal(ours[2], ref[2])


@pytest.mark.parametrize("K", [1, 5, 10, 16, 28, 60, 150])
@pytest.mark.parametrize("depths", ["uniform", "quantised", "one_depth", "uniform_5200"])
def test_points_lists_longer_than_one_sort_round(K, depths):
    """raster_points.hip: point_sorted_kernel sorts at most 768 candidates of a sub-tile per round (round 5).  One cloud of 3300
    points, every splat wider than the image: each sub-tile's list takes five rounds -- the first APPENDS to the queues, the
    later ones INSERT and drop what lies behind the wave's largest K-th key.  Depth distributions: uniform; quantised to eight
    values (a bucket holds ~100 exact ties, ordered by index); ONE depth for all (the bucket sort degenerates to the rank pass).
    A second, small cloud checks that the rounds of one image do not leak into the next."""
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(7000 + K)
    # 3300 points in the first cloud: every tile's list fits the tile-sorted kernel's LDS copy (2048 entries); 5100: it does not --
    # that kernel then inserts the hits of the unsorted tail, the single-wave sorted kernel (K > 28) takes seven rounds
    P = 5200 if depths == "uniform_5200" else 3400
    pts = _cloud(P, gen, zlo=0.1, zhi=2.0)
    if depths == "quantised":
        pts[:, 2] = (pts[:, 2] * 4).round() / 4 + 0.25
    elif depths == "one_depth":
        pts[:, 2] = 1.25
    pts[::97, 2] = -0.5  # behind the camera: dropped
    first = torch.tensor([0, P - 100])
    count = torch.tensor([P - 100, 100])
    radius = torch.rand(P, generator=gen) * 0.5 + 2.5
    size = (24, 40)
    ref = orc.rasterize_points_naive(pts, first, count, size, radius, K)
    for bin_size in (0, 8, 32):
        ours = _C.rasterize_points(pts.to(d), first.to(d), count.to(d), size, radius.to(d), K, bin_size, 6000)
        ours = [o.cpu() for o in ours]
        assert torch.equal(ours[0], ref[0]), f"idx differs K={K} bin={bin_size}: {(ours[0] != ref[0]).sum().item()}"
        assert torch.equal(ours[1], ref[1]) and torch.equal(ours[2], ref[2])


@pytest.mark.parametrize("size,K", [((1100, 900), 6), ((640, 2048), 10)])
def test_points_on_images_larger_than_the_bin_grid(size, K):
    """Above 512 pixels the internal bins grow beyond one tile: naive == binned (bit-exact), sorted K-prefixes, -1 padding."""
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(K)
    P = 60000
    pts = torch.cat([torch.rand(P, 2, generator=gen) * 2.4 - 1.2, torch.rand(P, 1, generator=gen) * 2 + 0.3], 1).to(d)
    first = torch.tensor([0, 25000], device=d)
    count = torch.tensor([25000, 35000], device=d)
    radius = (torch.rand(P, generator=gen) * 0.02 + 0.004).to(d)
    a = _C.rasterize_points(pts, first, count, size, radius, K, 128, 60000)
    b = _C.rasterize_points(pts, first, count, size, radius, K, 0, 0)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    idx, zbuf, dists = (t.cpu() for t in a)
    valid = idx >= 0
    assert (zbuf[~valid] == -1).all() and (dists[~valid] == -1).all()
    assert (valid[..., 1:] <= valid[..., :-1]).all()
    z = torch.where(valid, zbuf, torch.full_like(zbuf, float("inf")))
    assert (z[..., 1:] >= z[..., :-1]).all()
    assert 0.02 < valid.float().mean() < 0.98
    assert (idx[0][valid[0]] < 25000).all() and (idx[1][valid[1]] >= 25000).all()


def test_points_coarse_and_fine_ops():
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(3)
    P = 3000
    pts = _cloud(P, gen)
    first = torch.tensor([0, 1500])
    count = torch.tensor([1500, 1500])
    radius = torch.rand(P, generator=gen) * 0.05 + 0.01
    for (H, W), bs in [((32, 32), 8), ((64, 40), 8), ((30, 50), 5)]:
        ref, _ = orc.rasterize_points_coarse(pts, first, count, (H, W), radius, bs, 600)
        ours = _C._rasterize_points_coarse(pts.to(d), first.to(d), count.to(d), (H, W), radius.to(d), bs, 600).cpu()
        assert torch.equal(ours, ref)
        fine = _C._rasterize_points_fine(pts.to(d), ours.to(d), (H, W), radius.to(d), bs, 6)
        naive = orc.rasterize_points_naive(pts, first, count, (H, W), radius, 6)
        assert all(torch.equal(a.cpu(), b) for a, b in zip(fine, naive))



def test_one_image_scan_tail_is_race_free():
    """binning.hip: bin_scan_rows_tail_kernel (one image, more than 128 chunks: the workgroup that draws the last ticket runs the offsets
    scan on totals other workgroups -- on other XCDs -- have just written).  300k points, 336 x 336 (441 rows, 111 workgroups), the
    coarse operator 200 times: every run must give the bins of the first, and the first the oracle's."""
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(77)
    P = 300_000
    pts = _cloud(P, gen, zlo=0.1)
    first, count = torch.tensor([0]), torch.tensor([P])
    radius = torch.full((P,), 0.004)
    size, bin_size, M = (336, 336), 16, 1200
    ref = orc.rasterize_points_coarse(pts, first, count, size, radius, bin_size, M)[0]
    args = (pts.to(d), first.to(d), count.to(d), size, radius.to(d), bin_size, M)
    want = _C._rasterize_points_coarse(*args)
    assert torch.equal(want.cpu(), ref)
    for it in range(200):
        got = _C._rasterize_points_coarse(*args)
        assert torch.equal(got, want), f"run {it}: {(got != want).sum().item()} entries differ"


@pytest.mark.parametrize("size", [(40, 56), (45, 59)])  # whole / ragged 8x8 tiles of the backward kernel
def test_points_backward_and_autograd(size):
    import pytorch3d_amd as p3d
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(8)
    P = 2000
    pts = _cloud(P, gen, zlo=0.1)
    clouds = p3d.PackedPointclouds([pts[:900].to(d).requires_grad_(True), pts[900:].to(d).requires_grad_(True)])
    idx, zbuf, dists = p3d.rasterize_points(clouds, image_size=size, radius=0.05, points_per_pixel=5)
    gz = torch.randn(zbuf.shape, generator=gen)
    gd = torch.randn(dists.shape, generator=gen)
    ref = orc.rasterize_points_backward(pts, idx.cpu(), gz, gd, acc64=True)
    direct = _C.rasterize_points_backward(pts.to(d), idx, gz.to(d), gd.to(d)).cpu()
    scale = ref.abs().max().item()
    assert torch.allclose(direct, ref, rtol=1e-4, atol=5e-6 * max(scale, 1.0))
    # and through autograd, including the concatenation of the two clouds
    p1 = pts[:900].to(d).requires_grad_(True)
    p2 = pts[900:].to(d).requires_grad_(True)

    class _C2:  # packed container over a differentiable concat
        _N, _P, device = 2, 1100, d

        def points_packed(self):
            return torch.cat([p1, p2], 0)

        def cloud_to_packed_first_idx(self):
            return torch.tensor([0, 900], device=d)

        def num_points_per_cloud(self):
            return torch.tensor([900, 1100], device=d)

    out = p3d.rasterize_points(_C2(), image_size=size, radius=0.05, points_per_pixel=5)
    torch.autograd.backward([out[1], out[2]], [gz.to(d), gd.to(d)])
    got = torch.cat([p1.grad, p2.grad], 0).cpu()
    assert torch.allclose(got, ref, rtol=1e-4, atol=5e-6 * max(scale, 1.0))


@pytest.mark.parametrize("mode", ["alphacomposite", "weightedsumnorm", "weightedsum"])
# 17 / 24 / 32: the <., 32> tile kernels of the backward (composite_bwd_tile_kernel<0, 32> holds 12 registers in AGPRs:
# pytorch3d_amd/build.py: AGPR_KERNELS_TESTED points here); about one index in 300 is -1, the rows are all but dense
@pytest.mark.parametrize("K", [4, 10, 16, 17, 24, 32, 40])
@pytest.mark.parametrize("permuted", [False, True])
def test_compositors(mode, K, permuted):
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(K)
    N, C, P, H, W = 2, (5 if K != 10 else 9), 300, 13, 17  # C = 9: three channel passes through the table, the last partial
    feat = torch.rand(C, P, generator=gen)
    if permuted:  # what the renderers pass: the transposed view of (P, C) features (round 5: read through their strides) ...
        feat = torch.rand(P, C, generator=gen).t()
    if permuted:  # ... and permuted views of (N,H,W,K) tensors
        alphas = torch.rand(N, H, W, K, generator=gen).permute(0, 3, 1, 2)
        idx = torch.randint(-1, P, (N, H, W, K), generator=gen).permute(0, 3, 1, 2)
    else:
        alphas = torch.rand(N, K, H, W, generator=gen)
        idx = torch.randint(-1, P, (N, K, H, W), generator=gen)
    go = torch.randn(N, C, H, W, generator=gen)
    ref = orc.composite_forward(mode, feat, alphas, idx)
    fwd = getattr(_C, "accum_" + mode)(feat.to(d), alphas.to(d), idx.to(d)).cpu()
    assert torch.equal(fwd, ref), f"forward not bit-exact: {(fwd - ref).abs().max().item()}"
    rgf, rga = orc.composite_backward(mode, go, feat, alphas, idx)
    gf, ga = getattr(_C, "accum_" + mode + "_backward")(go.to(d), feat.to(d), alphas.to(d), idx.to(d))
    assert tuple(gf.shape) == (C, P) and (gf.stride() == ((1, C) if permuted else (P, 1)))  # the gradient in the features' layout
    assert torch.allclose(gf.cpu(), rgf, atol=2e-5, rtol=1e-4)
    assert torch.allclose(ga.cpu(), rga, atol=2e-5 * max(1.0, rga.abs().max().item()), rtol=1e-4)
    ref_mod = orc.ref_module()
    if ref_mod is not None:
        r = getattr(ref_mod, "accum_" + mode)(feat, alphas.contiguous(), idx.contiguous())
        assert torch.allclose(fwd, r, atol=1e-6)


def test_compositor_autograd_mirror():
    import pytorch3d_amd as p3d

    d = _dev()
    gen = torch.Generator().manual_seed(1)
    N, C, P, K, H, W = 1, 3, 100, 6, 9, 11
    feat = torch.rand(C, P, generator=gen).to(d).requires_grad_(True)
    alphas = torch.rand(N, K, H, W, generator=gen).to(d).requires_grad_(True)
    idx = torch.randint(-1, P, (N, K, H, W), generator=gen).to(d)
    for fn, mode in ((p3d.alpha_composite, "alphacomposite"), (p3d.norm_weighted_sum, "weightedsumnorm"),
                     (p3d.weighted_sum, "weightedsum")):
        feat.grad = alphas.grad = None
        img = fn(idx, alphas, feat)
        go = torch.randn(img.shape, generator=gen)
        img.backward(go.to(d))
        rgf, rga = orc.composite_backward(mode, go, feat.detach().cpu(), alphas.detach().cpu(), idx.cpu())
        assert torch.allclose(feat.grad.cpu(), rgf, atol=2e-5, rtol=1e-4)
        assert torch.allclose(alphas.grad.cpu(), rga, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("D", [1, 3, 4, 7, 8, 32])
def test_interp_face_attrs(D):
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(D)
    P, F = 5000, 60
    p2f = torch.randint(-1, F, (P,), generator=gen)
    bary = torch.rand(P, 3, generator=gen)
    attrs = torch.randn(F, 3, D, generator=gen)
    ref = orc.interp_forward(p2f, bary, attrs)
    out = _C.interp_face_attrs_forward(p2f.to(d), bary.to(d), attrs.to(d)).cpu()
    assert torch.equal(out, ref)
    g = torch.randn(P, D, generator=gen)
    rgb, rgf = orc.interp_backward(p2f, bary, attrs, g)
    gb, gf = _C.interp_face_attrs_backward(p2f.to(d), bary.to(d), attrs.to(d), g.to(d))
    assert torch.allclose(gb.cpu(), rgb, atol=1e-5, rtol=1e-5)
    assert torch.allclose(gf.cpu(), rgf, atol=1e-4, rtol=1e-4)
    # float64 is dispatched too (AT_DISPATCH_FLOATING_TYPES, interp_face_attrs.cu:72)
    out64 = _C.interp_face_attrs_forward(p2f.to(d), bary.double().to(d), attrs.double().to(d)).cpu()
    assert torch.allclose(out64.float(), ref, atol=1e-6)


def test_interp_mirror_shapes_and_grads():
    import pytorch3d_amd as p3d

    d = _dev()
    gen = torch.Generator().manual_seed(2)
    N, H, W, K, F, D = 2, 6, 7, 3, 20, 4
    p2f = torch.randint(-1, F, (N, H, W, K), generator=gen).to(d)
    bary = torch.rand(N, H, W, K, 3, generator=gen).to(d).requires_grad_(True)
    attrs = torch.randn(F, 3, D, generator=gen).to(d).requires_grad_(True)
    out = p3d.interpolate_face_attributes(p2f, bary, attrs)
    assert out.shape == (N, H, W, K, D)
    # python reference (pytorch3d/ops/interp_face_attrs.py:86-102 semantics) for values and grads
    mask = (p2f < 0)
    idx = p2f.clone()
    idx[mask] = 0
    pf = attrs[idx]  # (N,H,W,K,3,D)
    ref = (bary[..., None] * pf).sum(-2)
    ref = torch.where(mask[..., None], torch.zeros_like(ref), ref)
    assert torch.allclose(out, ref, atol=1e-6)
    g = torch.randn(out.shape, generator=gen).to(d)
    ga, gb = torch.autograd.grad(ref, [attrs, bary], g, retain_graph=True)
    oa, ob = torch.autograd.grad(out, [attrs, bary], g)
    assert torch.allclose(oa, ga, atol=1e-4) and torch.allclose(ob, gb, atol=1e-5)
    with pytest.raises(ValueError):
        p3d.interpolate_face_attributes(p2f[0], bary, attrs)


# ---------------------------------------------------------------------------------------------
# BASELINE full sizes: the oracle is too slow there, so check size-independent properties
# ---------------------------------------------------------------------------------------------
def test_config4_points_properties_at_full_size():
    """BASELINE configs[3]: 1M points, 512x512, K=10, r=0.01.  Two different binnings must agree bit for bit;
    every slot is sorted by (z, idx), points lie within r of the pixel centre, zbuf is the point's own z,
    the backward equals the closed form 2*g*(p - pix) accumulated per point (checked through a dense
    re-derivation from the forward outputs with torch ops)."""
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(0)
    P, H, W, K, r = 1_000_000, 512, 512, 10, 0.01
    pts = torch.cat([torch.rand(P, 2, generator=gen) * 2 - 1, torch.rand(P, 1, generator=gen) * 2 + 0.5], 1).to(d)
    first = torch.zeros(1, dtype=torch.int64, device=d)
    count = torch.full((1,), P, dtype=torch.int64, device=d)
    radius = torch.full((P,), r, device=d)
    a = _C.rasterize_points(pts, first, count, (H, W), radius, K, 32, 200000)
    b = _C.rasterize_points(pts, first, count, (H, W), radius, K, 64, 400000)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    idx, zbuf, dists = a
    valid = idx >= 0
    assert valid.float().mean() > 0.99  # ~78 points cover every pixel
    assert (dists[valid] < r * r).all() and (dists[valid] >= 0).all()
    li = idx.long().clamp(min=0)
    assert torch.equal(zbuf[valid], pts[:, 2][li][valid])
    z = torch.where(valid, zbuf, torch.full_like(zbuf, float("inf")))
    assert (z[..., 1:] >= z[..., :-1]).all()
    tie = (z[..., 1:] == z[..., :-1]) & valid[..., 1:]
    assert (idx[..., 1:][tie] > idx[..., :-1][tie]).all()
    # dists re-derived from the indices: pixel centres with the flipped axes (rasterize_points.cu:117-121)
    ys = torch.arange(H, device=d, dtype=torch.float32)
    xs = torch.arange(W, device=d, dtype=torch.float32)
    py = (1.0 - (2.0 * ys + 1.0) / H).view(1, H, 1, 1)
    px = (1.0 - (2.0 * xs + 1.0) / W).view(1, 1, W, 1)
    dx = pts[:, 0][li] - px
    dy = pts[:, 1][li] - py
    assert torch.allclose((dx * dx + dy * dy)[valid], dists[valid], atol=1e-6, rtol=0)
    # backward vs the closed form (rasterize_points.cu:402-409), accumulated with index_add in float64
    gz = torch.randn(idx.shape, generator=gen).to(d)
    gd = torch.randn(idx.shape, generator=gen).to(d)
    got = _C.rasterize_points_backward(pts, idx, gz, gd)
    ref = torch.zeros(P, 3, dtype=torch.float64, device=d)
    w = valid.double()
    ref[:, 0].index_add_(0, li.reshape(-1), (2.0 * gd.double() * dx.double() * w).reshape(-1))
    ref[:, 1].index_add_(0, li.reshape(-1), (2.0 * gd.double() * dy.double() * w).reshape(-1))
    ref[:, 2].index_add_(0, li.reshape(-1), (gz.double() * w).reshape(-1))
    assert torch.allclose(got.double(), ref, atol=5e-5, rtol=1e-4)


@pytest.mark.parametrize("mode", ["alphacomposite", "weightedsumnorm", "weightedsum"])
def test_config4_compositor_properties_at_full_size(mode):
    """512x512, K=10, P=1M, C=3 on permuted (N,H,W,K) views: linear in the features; forward equals a
    dense torch re-derivation; backward equals torch autograd of that re-derivation."""
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(1)
    N, H, W, K, P, C = 1, 512, 512, 10, 1_000_000, 3
    idx = torch.randint(-1, P, (N, H, W, K), generator=gen).to(d).permute(0, 3, 1, 2)
    # alphas < 0.95: the reference's alpha backward divides by (1 - alpha + 1e-9) (alpha_composite.cu:20,135), which
    # only equals the analytic derivative away from alpha = 1
    alphas = (torch.rand(N, H, W, K, generator=gen) * 0.95).to(d).permute(0, 3, 1, 2)
    f1 = torch.rand(C, P, generator=gen).to(d)
    f2 = torch.rand(C, P, generator=gen).to(d)
    fwd = getattr(_C, "accum_" + mode)
    o1, o2, o12 = fwd(f1, alphas, idx), fwd(f2, alphas, idx), fwd(f1 + f2, alphas, idx)
    assert torch.allclose(o1 + o2, o12, atol=1e-5, rtol=1e-5)

    def dense(feats, al):
        ok = (idx >= 0).float()
        g = feats[:, idx.clamp(min=0)]  # (C, N, K, H, W)
        a = al * ok
        if mode == "alphacomposite":
            one_minus = torch.where(idx >= 0, 1 - al, torch.ones_like(al))
            cum = torch.cumprod(torch.cat([torch.ones_like(al[:, :1]), one_minus[:, :-1]], 1), 1)
            wgt = a * cum
        elif mode == "weightedsumnorm":
            wgt = a / a.sum(1, keepdim=True).clamp(min=1e-4)
        else:
            wgt = a
        return (g * wgt.unsqueeze(0)).sum(2).permute(1, 0, 2, 3)

    fr = f1.clone().requires_grad_(True)
    ar = alphas.clone().requires_grad_(True)
    ref = dense(fr, ar)
    assert torch.allclose(o1, ref, atol=2e-5, rtol=1e-5)
    go = torch.randn(ref.shape, generator=gen).to(d)
    ref.backward(go)
    gf, ga = getattr(_C, "accum_" + mode + "_backward")(go, f1, alphas, idx)
    assert torch.allclose(gf, fr.grad, atol=1e-4, rtol=1e-4)
    assert torch.allclose(ga, ar.grad, atol=1e-4, rtol=1e-3)


def test_interp_properties_at_full_fragment_size():
    """P = 16 x 512 x 512 x 8 = 33.5M samples (a quarter of the bench fragments), D = 3: forward equals the torch
    gather formula (interp_face_attrs.py:86-102), backward equals its autograd."""
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(2)
    Pn, F, D = 16 * 512 * 512 * 8, 100_000, 3
    # runs of equal faces, like real fragments, with 60% background
    base = torch.randint(0, F, (Pn // 16,), generator=gen).repeat_interleave(16)
    p2f = torch.where(torch.rand(Pn, generator=gen) < 0.6, torch.full((Pn,), -1), base).to(d)
    bary = torch.rand(Pn, 3, generator=gen).to(d)
    attrs = torch.randn(F, 3, D, generator=gen).to(d)
    out = _C.interp_face_attrs_forward(p2f, bary, attrs)
    br = bary.clone().requires_grad_(True)
    ar = attrs.clone().requires_grad_(True)
    ok = (p2f >= 0).float().view(-1, 1)
    ref = (br.unsqueeze(-1) * ar[p2f.clamp(min=0)]).sum(1) * ok
    assert torch.allclose(out, ref, atol=1e-5, rtol=1e-5)
    g = torch.randn(Pn, D, generator=gen).to(d)
    ref.backward(g)
    gb, ga = _C.interp_face_attrs_backward(p2f, bary, attrs, g)
    assert torch.allclose(gb, br.grad, atol=1e-5, rtol=1e-5)
    assert torch.allclose(ga, ar.grad, atol=2e-3, rtol=2e-3)
    # the image-shaped variant (what pytorch3d_amd.interpolate_face_attributes uses): same numbers
    for shape in ((16, 512, 512, 8), (8, 512, 1024, 8), (64, 512, 512, 2), (1, 4096, 4096, 2), (32, 512, 512, 4)):
        gb2, ga2 = _C.interp_face_attrs_backward(p2f, bary, attrs, g, image_shape=shape)
        assert torch.allclose(gb2, br.grad, atol=1e-5, rtol=1e-5), shape
        assert torch.allclose(ga2, ar.grad, atol=2e-3, rtol=2e-3), shape
    for D2 in (1, 2, 4):  # generic-K path (K = 3) and the other D instantiations, odd image sizes
        n, h, w, k = 3, 37, 53, 3
        P2 = n * h * w * k
        p2 = p2f[:P2].clone()
        b2 = bary[:P2].clone().requires_grad_(True)
        a2 = torch.randn(F, 3, D2, generator=gen).to(d).requires_grad_(True)
        g2 = torch.randn(P2, D2, generator=gen).to(d)
        ((b2.unsqueeze(-1) * a2[p2.clamp(min=0)]).sum(1) * (p2 >= 0).float().view(-1, 1) * g2).sum().backward()
        gb3, ga3 = _C.interp_face_attrs_backward(p2, b2.detach(), a2.detach(), g2, image_shape=(n, h, w, k))
        assert torch.allclose(gb3, b2.grad, atol=1e-5, rtol=1e-5), D2
        assert torch.allclose(ga3, a2.grad, atol=1e-4, rtol=1e-4), D2
