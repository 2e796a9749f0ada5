"""TEST INFRASTRUCTURE (oracle): the reference's mesh-rasterization backward in float64, per sample, vectorised in torch.

    grad, abs_sum = backward_f64(face_verts, pix_to_face, grad_zbuf, grad_bary, grad_dists, persp, clip)

Restates RasterizeMeshesBackwardCudaKernel (pytorch3d/csrc/rasterize_meshes/rasterize_meshes.cu:433-564) with the math of
pytorch3d/csrc/utils/geometry_utils.cuh in float64 on whatever device the tensors live on, sample by sample:

  * `grad`     (F,3,3) f64  the sum over a face's (pixel, k) samples of the nine partials;
  * `abs_sum`  (F,3,3) f64  the sum of their ABSOLUTE values: the scale against which a float32 sum of the same terms in
                            any order (float atomics) can be judged -- |error| <= c * eps * abs_sum.

Why it exists (round 3).  At the bench size the batch holds faces seen nearly edge-on.  In the blur band far outside such a
face the perspective denominator `bw.x z1 z2 + z0 bw.y z2 + z0 z1 bw.z` (geometry_utils.cuh:172-185) goes negative and
is clamped at 1e-8, and the backward multiplies by 1 / denom^2 = 1e16 (geometry_utils.cuh:214).  Where ONE barycentric
survives the clipping the clipped coordinates are the constant (0, 0, 1) and the true gradient through them is 0; the
reference gets there by the difference `1 / s - w / s^2` of two rounded float quotients (geometry_utils.cuh:313-327), which
is 0 or one ulp depending on how they round -- times 1e16.  Its CPU build, its device build and a C restatement with
`s * s` for `pow(s, 2)` disagree with each other on those samples (0 vs 1e15), so there is no reference VALUE to compare
with; what can be checked is the mathematical one.  This module evaluates the reference's formulas (CUDA semantics: the
clip backward is fed the PRE-perspective barycentrics, rasterize_meshes.cu:528 -- a quirk, kept) in float64 with that
one term written cancellation-free, `(w_j + w_k) / s^2`, which is algebraically the same expression.

Pinned by tests/test_cpu_oracle_golden.py against oracle/p3d_oracle.c (float32, the reference's operation order) on
well-conditioned soups.  Only tests/ import this.
"""
import torch

KEPS = 1e-8  # geometry_utils.cuh:18


def _ndc(i, S1, S2):  # rasterization_utils.cuh:16-42, float32 like the kernels (the pixel centre is an INPUT of the math)
    rng = torch.tensor(2.0, dtype=torch.float32)
    if S1 > S2:
        rng = (torch.tensor(float(S1), dtype=torch.float32) * rng) / torch.tensor(float(S2), dtype=torch.float32)
    rng = rng.to(i.device)
    off = rng / 2.0
    return (-off + (rng * i.to(torch.float32) + off) / float(S1)).to(torch.float64)


def _edge(p, a, b):  # geometry_utils.cuh:37-41
    return (p[..., 0] - a[..., 0]) * (b[..., 1] - a[..., 1]) - (p[..., 1] - a[..., 1]) * (b[..., 0] - a[..., 0])


def _edge_bwd(p, a, b, g):  # geometry_utils.cuh:54-64 -> (dp, da, db), each (M,2)
    dp = torch.stack([g * (b[..., 1] - a[..., 1]), g * (a[..., 0] - b[..., 0])], -1)
    da = torch.stack([g * (p[..., 1] - b[..., 1]), g * (b[..., 0] - p[..., 0])], -1)
    db = torch.stack([g * (a[..., 1] - p[..., 1]), g * (p[..., 0] - a[..., 0])], -1)
    return dp, da, db


def _seg_dist2(p, a, b):  # geometry_utils.cuh:340-352
    ba = b - a
    l2 = (ba * ba).sum(-1)
    t = ((ba * (p - a)).sum(-1) / torch.where(l2 > 0, l2, torch.ones_like(l2))).clamp(0.0, 1.0)
    proj = a + t[..., None] * ba
    d_seg = ((proj - p) ** 2).sum(-1)
    d_pt = ((p - b) ** 2).sum(-1)
    return torch.where(l2 <= KEPS, d_pt, d_seg)


def _seg_dist2_bwd(p, a, b, g):  # geometry_utils.cuh:365-385 -> (da, db)
    ba = b - a
    bot = (ba * ba).sum(-1)
    top = (ba * (p - a)).sum(-1)
    tt = (top / torch.where(bot != 0, bot, torch.ones_like(bot))).clamp(0.0, 1.0)
    proj = (1.0 - tt)[..., None] * a + tt[..., None] * b
    d = proj - p
    return (g * (1.0 - tt) * 2.0)[..., None] * d, (g * tt * 2.0)[..., None] * d


def per_sample_backward_f64(fv, p, gz, gb, gd, persp, clip):
    """fv (M,3,3) f64 the sample's face, p (M,2) f64 pixel centre, gz (M,), gb (M,3), gd (M,) f64 -> (M,3,3) f64."""
    a, b, c = fv[:, 0, :2], fv[:, 1, :2], fv[:, 2, :2]
    z0, z1, z2 = fv[:, 0, 2], fv[:, 1, 2], fv[:, 2, 2]
    area = _edge(c, a, b) + KEPS  # geometry_utils.cuh:76-79
    e0, e1, e2 = _edge(p, b, c), _edge(p, c, a), _edge(p, a, b)
    bw = torch.stack([e0 / area, e1 / area, e2 / area], -1)
    bp = bw
    if persp:  # geometry_utils.cuh:172-185 (CUDA product order is irrelevant in f64)
        t = torch.stack([bw[:, 0] * z1 * z2, z0 * bw[:, 1] * z2, z0 * z1 * bw[:, 2]], -1)
        den = t.sum(-1).clamp_min(KEPS)
        bp = t / den[:, None]
    bc = bp
    if clip:  # geometry_utils.cuh:246-259
        w = bp.clamp_min(0.0)
        bc = w / w.sum(-1).clamp_min(1e-5)[:, None]
    inside = (bp > 0).all(-1)
    sign = torch.where(inside, -torch.ones_like(gd), torch.ones_like(gd))

    # PointTriangleDistanceBackward (geometry_utils.cuh:421-462): the closest edge, ties e01, e02, e12
    d01, d02, d12 = _seg_dist2(p, a, b), _seg_dist2(p, a, c), _seg_dist2(p, b, c)
    s0 = (d01 <= d02) & (d01 <= d12)
    s1 = ~s0 & (d02 <= d01) & (d02 <= d12)
    s2 = ~s0 & ~s1 & (d12 <= d01) & (d12 <= d02)
    ea = torch.where(s2[:, None], b, a)
    eb = torch.where(s0[:, None], b, c)
    g_d = torch.where(s0 | s1 | s2, sign * gd, torch.zeros_like(gd))
    da, db = _seg_dist2_bwd(p, ea, eb, g_d)
    zero = torch.zeros_like(da)
    dd0 = torch.where((s0 | s1)[:, None], da, zero)
    dd1 = torch.where(s0[:, None], db, torch.where(s2[:, None], da, zero))
    dd2 = torch.where((s1 | s2)[:, None], db, zero)

    g = torch.stack([gb[:, 0] + gz * z0, gb[:, 1] + gz * z1, gb[:, 2] + gz * z2], -1)  # rasterize_meshes.cu:520-523
    if clip:  # BarycentricClipBackward on the PRE-perspective barycentrics (rasterize_meshes.cu:528)
        w = bw.clamp_min(0.0)
        s = w.sum(-1)
        live = s >= 1e-5
        s = torch.where(live, s, torch.full_like(s, 1e-5))
        inv_s2 = torch.where(live, 1.0 / (s * s), torch.zeros_like(s))
        others = torch.stack([w[:, 1] + w[:, 2], w[:, 0] + w[:, 2], w[:, 0] + w[:, 1]], -1)
        own = torch.where(live[:, None], others * inv_s2[:, None], (1.0 / s)[:, None].expand(-1, 3))  # = 1/s - w_k/s^2
        q = -w * inv_s2[:, None]
        cross = (g * q).sum(-1, keepdim=True) - g * q
        g = torch.where(bw < 0, torch.zeros_like(g), g * own + cross)
    dz = torch.zeros_like(g)
    if persp:  # BarycentricPerspectiveCorrectionBackward (geometry_utils.cuh:200-228)
        t = torch.stack([bw[:, 0] * z1 * z2, z0 * bw[:, 1] * z2, z0 * z1 * bw[:, 2]], -1)
        den = t.sum(-1).clamp_min(KEPS)
        g_den = -(t * g).sum(-1) / (den * den)
        gt = g_den[:, None] + g / den[:, None]
        gnew = torch.stack([gt[:, 0] * z1 * z2, gt[:, 1] * z0 * z2, gt[:, 2] * z0 * z1], -1)
        dz = torch.stack([gt[:, 1] * bw[:, 1] * z2 + gt[:, 2] * bw[:, 2] * z1, gt[:, 0] * bw[:, 0] * z2 + gt[:, 2] * bw[:, 2] * z0,
                          gt[:, 0] * bw[:, 0] * z1 + gt[:, 1] * bw[:, 1] * z0], -1)
        g = gnew
    # BarycentricCoordsBackward (geometry_utils.cuh:101-161)
    area2 = area * area
    out = [torch.zeros_like(a) for _ in range(3)]
    for k, (ek, (u, v), (iu, iv)) in enumerate(((e0, (b, c), (1, 2)), (e1, (c, a), (2, 0)), (e2, (a, b), (0, 1)))):
        _dp, du, dv = _edge_bwd(p, u, v, g[:, k] / area)
        out[iu] = out[iu] + du
        out[iv] = out[iv] + dv
        ap, aa, ab = _edge_bwd(c, a, b, g[:, k] * (-ek / area2))  # area = edge(v2, v0, v1): dp -> v2, da -> v0, db -> v1
        out[2] = out[2] + ap
        out[0] = out[0] + aa
        out[1] = out[1] + ab
    res = torch.zeros((fv.shape[0], 3, 3), dtype=torch.float64, device=fv.device)
    res[:, 0, :2] = out[0] + dd0
    res[:, 1, :2] = out[1] + dd1
    res[:, 2, :2] = out[2] + dd2
    res[:, :, 2] = gz[:, None] * bc + dz
    return res


def backward_f64(face_verts, pix_to_face, grad_zbuf, grad_bary, grad_dists, persp, clip, chunk=4_000_000):
    """-> (grad (F,3,3) f64, abs_sum (F,3,3) f64), see the module docstring."""
    N, H, W, K = pix_to_face.shape
    F = face_verts.shape[0]
    dev = face_verts.device
    grad = torch.zeros((F, 3, 3), dtype=torch.float64, device=dev)
    abs_sum = torch.zeros((F, 3, 3), dtype=torch.float64, device=dev)
    idx = (pix_to_face >= 0).nonzero()
    fvd = face_verts.to(torch.float64)
    for s in range(0, idx.shape[0], chunk):
        i = idx[s:s + chunk]
        n, yo, xo, k = i[:, 0], i[:, 1], i[:, 2], i[:, 3]
        f = pix_to_face[n, yo, xo, k]
        p = torch.stack([_ndc(W - 1 - xo, W, H), _ndc(H - 1 - yo, H, W)], -1)  # rasterize_meshes.cu:458-462
        g = per_sample_backward_f64(fvd[f], p, grad_zbuf[n, yo, xo, k].double(), grad_bary[n, yo, xo, k].double(),
                                    grad_dists[n, yo, xo, k].double(), persp, clip)
        grad.index_add_(0, f, g)
        abs_sum.index_add_(0, f, g.abs())
    return grad, abs_sum
