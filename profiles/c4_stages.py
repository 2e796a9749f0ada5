import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import pytorch3d_amd as p3d
from pytorch3d_amd import _C
d = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
P = 1_000_000
pts = torch.cat([torch.rand(P, 2, generator=gen) * 2 - 1, torch.rand(P, 1, generator=gen) * 2 + 0.5], 1).to(d)
feats = torch.rand(P, 3, generator=gen).to(d)
H = W = 512; K = 10; r = 0.01
pts_g = pts.clone().requires_grad_(True)
feats_g = feats.clone().requires_grad_(True)
g_img = torch.randn(1, 3, H, W, generator=gen).to(d)
pc = p3d.PackedPointclouds([pts_g])
def T(name, fn, it=5):
    for _ in range(2): out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): out = fn()
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter()-t0)/it*1e3:.3f} ms", flush=True)
    return out
from pytorch3d_amd.rasterize_points import _format_radius
rad = T("format_radius", lambda: _format_radius(r, pc))
first, cnt = pc.cloud_to_packed_first_idx(), pc.num_points_per_cloud()
T("accessors", lambda: (pc.points_packed(), pc.cloud_to_packed_first_idx(), pc.num_points_per_cloud()))
out = T("_C.rasterize_points", lambda: _C.rasterize_points(pts, first, cnt, (H, W), rad, K, 32, 200000))
idx, zbuf, dists = out
T("p3d.rasterize_points(nograd pc)", lambda: p3d.rasterize_points(p3d.PackedPointclouds([pts]), image_size=H, radius=r, points_per_pixel=K))
T("p3d.rasterize_points(grad)", lambda: p3d.rasterize_points(pc, image_size=H, radius=r, points_per_pixel=K))
gz = torch.randn_like(zbuf); gd = torch.randn_like(dists)
T("_C.rasterize_points_backward", lambda: _C.rasterize_points_backward(pts, idx, gz, gd))
w = (1 - dists.permute(0, 3, 1, 2) / (r * r))
il = idx.long().permute(0, 3, 1, 2)
T("weights+long", lambda: ((1 - dists.permute(0, 3, 1, 2) / (r * r)), idx.long().permute(0, 3, 1, 2)))
ft = feats.permute(1, 0)
img = T("_C.accum_alphacomposite", lambda: _C.accum_alphacomposite(ft, w, il))
T("_C.accum_alphacomposite_backward", lambda: _C.accum_alphacomposite_backward(g_img, ft, w, il))
def full():
    pts_g.grad = None; feats_g.grad = None
    idx, zbuf, dists = p3d.rasterize_points(pc, image_size=H, radius=r, points_per_pixel=K)
    weights = 1 - dists.permute(0, 3, 1, 2) / (r * r)
    img = p3d.alpha_composite(idx.long().permute(0, 3, 1, 2), weights, feats_g.permute(1, 0))
    return img
img = T("fwd only (autograd graph)", full)
def fb():
    img = full(); img.backward(g_img)
T("fwd+bwd", fb)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): fb()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
