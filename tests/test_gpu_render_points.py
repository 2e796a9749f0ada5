"""GPU parity: PointsRenderer's chain as two launches (include/p3d_amd.h: p3d_rasterize_points_composite, _composite_backward;
pytorch3d_amd.render_points) against
  * the C oracle's rasterize_points_naive + composite_forward on `weights = 1 - dists / r^2` (renderer/points/renderer.py:56-76):
    fragments and image bit-exact;
  * the package's own operators run one after the other (each pinned to the oracle in test_gpu_points_composite_interp.py): image
    bit-equal, gradients within 1e-4 of their largest entry (the gate of tests/test_gpu_points_renderer_dropin.py);
  * a float64 torch restatement of the chain on the SAME fragments (autograd through the gathered points): gradients within 2e-4 of
    their largest entry (float32 sums of ~K x pi r^2 terms per point).
"""
import pytest
import torch

import _util as U  # noqa: F401
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _cloud(P, gen, zlo=-0.2, zhi=2.0):
    return torch.cat([torch.rand(P, 2, generator=gen) * 2.4 - 1.2, torch.rand(P, 1, generator=gen) * (zhi - zlo) + zlo], 1)


_MODES = {"alpha": "alphacomposite", "norm": "weightedsumnorm"}  # fused node's name -> the oracle's


def _compositor(mode):
    from pytorch3d_amd import compositing

    return compositing.alpha_composite if mode == "alpha" else compositing.norm_weighted_sum


def _chain(pts, first, count, size, radius_t, feats, r, K, bin_size, cap, mode="alpha"):
    """The operators one after the other, as PointsRenderer.forward writes them."""
    from pytorch3d_amd import _C

    idx, zbuf, dists = _C.rasterize_points(pts, first, count, size, radius_t, K, bin_size, cap)
    weights = 1 - dists.permute(0, 3, 1, 2) / (r * r)
    images = _compositor(mode)(idx.long().permute(0, 3, 1, 2), weights, feats.permute(1, 0))
    return images.permute(0, 2, 3, 1), idx, zbuf, dists


@pytest.mark.parametrize("mode", ["alpha", "norm"])
@pytest.mark.parametrize("size", [(32, 32), (20, 48), (45, 31)])
@pytest.mark.parametrize("K,C", [(1, 3), (4, 1), (5, 2), (10, 3), (16, 4), (20, 3), (28, 3), (40, 3)])
def test_fused_forward_vs_oracle_and_operator_chain(size, K, C, mode):
    """Every launch path of the fused entry: the tile-sorted kernel's epilogue (binned, K <= 28), the pass behind the single-wave
    sorted kernel (K > 28) and behind the naive launch (bin_size 0); ragged clouds (one empty), images with partial tiles."""
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(K * 100 + size[0] + C)
    P, r = 900, 0.2
    pts = _cloud(P, gen)
    first = torch.tensor([0, 350, 350])
    count = torch.tensor([350, 0, 550])
    feats = torch.rand(P, C, generator=gen)
    radius_t = torch.full((P,), r)
    ref = orc.rasterize_points_naive(pts, first, count, size, radius_t, K)
    # weights as torch evaluates `1 - dists / (r * r)` on float32 tensors (CPU and GPU agree: a multiplication by float(1 / (r r)))
    inv = _C.inv_r2_of(r)
    weights = 1 - ref[2].permute(0, 3, 1, 2) * torch.tensor(inv, dtype=torch.float32)
    want = orc.composite_forward(_MODES[mode], feats.t().contiguous(), weights.contiguous(), ref[0].long().permute(0, 3, 1, 2).contiguous())
    want = want.permute(0, 2, 3, 1)
    for bin_size in (0, 8, 16):
        idx, zbuf, dists, img = _C.rasterize_points_composite(pts.to(d), first.to(d), count.to(d), size, radius_t.to(d), feats.to(d), inv, K,
                                                              bin_size, 600, mode)
        assert torch.equal(idx.cpu(), ref[0]) and torch.equal(zbuf.cpu(), ref[1]) and torch.equal(dists.cpu(), ref[2]), f"fragments bin={bin_size}"
        assert torch.equal(img.cpu(), want), f"image vs oracle bin={bin_size}: {(img.cpu() - want).abs().max().item()}"
        chain = _chain(pts.to(d), first.to(d), count.to(d), size, radius_t.to(d), feats.to(d), r, K, bin_size, 600, mode)
        assert torch.equal(img, chain[0]), f"image vs the operator chain bin={bin_size}: {(img - chain[0]).abs().max().item()}"


def test_fused_forward_edge_cases():
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(5)
    z64 = lambda *v: torch.tensor(v, dtype=torch.int64, device=d)
    # no points at all: empty fragments, black image
    idx, zbuf, dists, img = _C.rasterize_points_composite(torch.zeros((0, 3), device=d), z64(0), z64(0), (16, 24), torch.zeros((0,), device=d),
                                                          torch.zeros((0, 3), device=d), 1.0, 4, 8, 100)
    assert tuple(img.shape) == (1, 16, 24, 3) and float(img.abs().max()) == 0.0 and int((idx != -1).sum()) == 0
    # every point behind the camera / off the image
    pts = _cloud(50, gen)
    pts[:, 2] = -1.0
    idx, _, _, img = _C.rasterize_points_composite(pts.to(d), z64(0), z64(50), (16, 24), torch.full((50,), 0.2, device=d),
                                                   torch.rand(50, 3, generator=gen).to(d), _C.inv_r2_of(0.2), 4, 8, 100)
    assert float(img.abs().max()) == 0.0 and int((idx != -1).sum()) == 0
    # channel counts the fused entries do not take
    with pytest.raises(RuntimeError, match="C in 1..4"):
        _C.rasterize_points_composite(pts.to(d), z64(0), z64(50), (16, 24), torch.full((50,), 0.2, device=d), torch.rand(50, 5).to(d), 1.0, 4, 8, 100)


def _f64_chain_grads(pts, feats, idx, size, r, g_img, mode="alpha"):
    """float64 torch autograd of the chain on FIXED fragments: dist2 from the gathered points and the pixel centres (rasterize_points.cu:
    55-60, un-flipped as :389-393), alpha = 1 - dist2 / r^2, alpha compositing front to back."""
    N, H, W, K = idx.shape
    p = pts.double().clone().requires_grad_(True)
    f = feats.double().clone().requires_grad_(True)

    def centres(S, other):  # pix_to_ndc of the stored (flipped) pixel index i: the centre of pixel S - 1 - i
        i = torch.arange(S, dtype=torch.float64)
        j = S - 1 - i
        if S >= other:
            rng = S / other
            return -rng + (2 * rng * j + rng) / S
        return -1 + (2 * j + 1) / S

    yf = centres(H, W)[None, :, None, None].expand(N, H, W, K)
    xf = centres(W, H)[None, None, :, None].expand(N, H, W, K)
    valid = idx >= 0
    ii = idx.long().clamp_min(0)
    q = p[ii]  # (N,H,W,K,3)
    d2 = (q[..., 0] - xf) ** 2 + (q[..., 1] - yf) ** 2
    al = torch.where(valid, 1 - d2 / (r * r), torch.zeros_like(d2))
    one_minus = torch.where(valid, 1 - al, torch.ones_like(al))
    if mode == "alpha":
        cum = torch.cumprod(torch.cat([torch.ones_like(al[..., :1]), one_minus[..., :-1]], -1), -1)
        wgt = torch.where(valid, cum * al, torch.zeros_like(al))
    else:  # norm_weighted_sum.cu:47-64: weights over max(their sum, 1e-4)
        wgt = al / al.sum(-1, keepdim=True).clamp_min(1e-4)
    img = (wgt[..., None] * f[ii]).sum(3)
    (img * g_img.double()).sum().backward()
    return img.detach(), p.grad, f.grad


@pytest.mark.parametrize("mode", ["alpha", "norm"])
@pytest.mark.parametrize("K,C", [(1, 3), (4, 2), (8, 4), (10, 3), (12, 1), (16, 3)])
@pytest.mark.parametrize("size", [(40, 56), (33, 21)])
def test_fused_backward_vs_operator_chain_and_float64(K, C, size, mode):
    from pytorch3d_amd import PackedPointclouds, render_points_alpha

    d = _dev()
    gen = torch.Generator().manual_seed(K * 10 + C + size[0])
    r = 0.25
    clouds = [_cloud(400, gen, zlo=0.1), _cloud(650, gen, zlo=0.1)]
    fl = [torch.rand(c.shape[0], C, generator=gen) for c in clouds]
    pts = torch.cat(clouds).to(d).requires_grad_(True)
    feats = torch.cat(fl).to(d).requires_grad_(True)
    pc = PackedPointclouds([pts[:400], pts[400:]])
    g_img = torch.randn((2,) + size + (C,), generator=gen)
    keep = torch.ones((2,) + size, dtype=torch.bool)
    if mode == "norm":
        # A pixel whose weights sum to less than ~1e-2 is divided by a number made of float32 rounding (1 - d / r^2 next to the splat's rim,
        # clamped at 1e-4: norm_weighted_sum.cu:55): float32 and float64 evaluations part there by construction.  Such pixels get no
        # upstream gradient and are left out of the image comparison with the float64 restatement (the operator chain is compared everywhere).
        from pytorch3d_amd import _C
        from pytorch3d_amd.rasterize_points import rasterize_points

        with torch.no_grad():
            i0, _, d0 = rasterize_points(pc, image_size=size, radius=r, points_per_pixel=K, bin_size=8, max_points_per_bin=700)
        asum = torch.where(i0 >= 0, 1 - d0 * _C.inv_r2_of(r), torch.zeros_like(d0)).sum(-1)
        keep = ((asum > 0.05) | (i0[..., 0] < 0)).cpu()
        g_img = g_img * keep[..., None]
    img, idx, zbuf, dists = render_points_alpha(pc, feats, image_size=size, radius=r, points_per_pixel=K, bin_size=8, max_points_per_bin=700,
                                                compositor=mode)
    (img * g_img.to(d)).sum().backward()
    gp, gf = pts.grad.clone(), feats.grad.clone()
    assert float(gp[:, 2].abs().max()) == 0.0  # the chain does not use zbuf
    assert int((idx[..., K - 1] >= 0).sum()) > 0, "no pixel fills its K slots"

    # the operators one after the other under autograd
    p2 = pts.detach().clone().requires_grad_(True)
    f2 = feats.detach().clone().requires_grad_(True)
    from pytorch3d_amd.rasterize_points import rasterize_points

    pc2 = PackedPointclouds([p2[:400], p2[400:]])
    idx2, _, d2 = rasterize_points(pc2, image_size=size, radius=r, points_per_pixel=K, bin_size=8, max_points_per_bin=700)
    img2 = _compositor(mode)(idx2.long().permute(0, 3, 1, 2), 1 - d2.permute(0, 3, 1, 2) / (r * r), f2.permute(1, 0)).permute(0, 2, 3, 1)
    assert torch.equal(idx, idx2) and torch.equal(img, img2)
    (img2 * g_img.to(d)).sum().backward()
    for name, a, b in (("points", gp, p2.grad), ("features", gf, f2.grad)):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 1e-4 * scale, (name, float((a - b).abs().max()), scale)

    # float64 restatement on the same fragments
    img64, gp64, gf64 = _f64_chain_grads(pts.detach().cpu(), feats.detach().cpu(), idx.cpu(), size, r, g_img, mode)
    assert float(((img.detach().cpu().double() - img64).abs() * keep[..., None]).max()) <= 1e-5
    for name, a, b in (("points", gp.cpu().double(), gp64), ("features", gf.cpu().double(), gf64)):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-4 * scale + 1e-6, (name, float((a - b).abs().max()), scale)  # (K = 1, norm: the image is the feature itself)


def test_fused_backward_refuses_what_it_does_not_take():
    from pytorch3d_amd import _C

    d = _dev()
    pts = torch.rand(10, 3, device=d)
    idx = torch.full((1, 8, 8, 20), -1, dtype=torch.int32, device=d)
    dists = torch.full((1, 8, 8, 20), -1.0, device=d)
    with pytest.raises(RuntimeError):
        _C.rasterize_points_composite_backward(pts, torch.rand(10, 3, device=d), idx, dists, torch.zeros(1, 8, 8, 3, device=d), 1.0)  # K > 16


@pytest.mark.parametrize("K", [4, 10, 40])
def test_fused_forward_with_short_workspaces(K):
    """include/p3d_amd.h "Short workspaces": the lists of the binned launch sized from a guess.  With lists that do NOT fit the naive
    kernel writes the fragments and a pass behind it, gated by the same device flag, the image; with lists that fit (the call after) the
    binned kernel writes both (K <= 28: its epilogue; the gated pass returns at once) -- same fragments, same image either way."""
    from pytorch3d_amd import _C

    d = _dev()
    gen = torch.Generator().manual_seed(31 + K)
    P, r, size = 1500, 0.15, (48, 80)
    pts = _cloud(P, gen).to(d)
    first, count = torch.tensor([0, 600], device=d), torch.tensor([600, 900], device=d)
    feats = torch.rand(P, 3, generator=gen).to(d)
    radius_t = torch.full((P,), r, device=d)
    inv = _C.inv_r2_of(r)
    saved = (_C.SHORT_WORKSPACE, _C.SHORT_WORKSPACE_FIRST_GUESS)
    try:
        _C.SHORT_WORKSPACE = "never"
        want = _C.rasterize_points_composite(pts, first, count, size, radius_t, feats, inv, K, 16, 1000)
        _C.SHORT_WORKSPACE, _C.SHORT_WORKSPACE_FIRST_GUESS = "always", 1
        _C._NEEDS.clear()
        for what in ("lists do not fit", "learned size"):
            got = _C.rasterize_points_composite(pts, first, count, size, radius_t, feats, inv, K, 16, 1000)
            assert _C.WORKSPACE_STATS["last_entries"] is not None
            for name, a, b in zip(("idx", "zbuf", "dists", "image"), got, want):
                assert torch.equal(a, b), (what, name)
            torch.cuda.synchronize()
    finally:
        _C.SHORT_WORKSPACE, _C.SHORT_WORKSPACE_FIRST_GUESS = saved
        _C._NEEDS.clear()
