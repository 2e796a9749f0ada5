import math, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util as U
from oracle import oracle as orc
import pytorch3d_amd as p3d
from pytorch3d_amd import _C
mod = orc.ref_hip_module(nofma=True)
d = torch.device("cuda:0")
B, H, K = 64, 512, 8
blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
verts, faces = U.hetero_batch(B, seed=0)
m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
fv = m.verts_packed()[m.faces_packed()].contiguous()
F = fv.shape[0]
first, count = m.mesh_to_faces_packed_first_idx(), m.num_faces_per_mesh()
nbr = torch.full((F,), -1, dtype=torch.int64, device=d)
ours = _C.rasterize_meshes(fv, first, count, nbr, (H, H), blur, K, 32, 64238, True, True, False)
gen = torch.Generator().manual_seed(231)
gz = torch.randn((B, H, H, K), generator=gen).to(d)
gb = torch.randn((B, H, H, K, 3), generator=gen).to(d)
gd = torch.randn((B, H, H, K), generator=gen).to(d)
a = _C.rasterize_meshes_backward(fv, ours[0], gz, gb, gd, True, True)
a2 = _C.rasterize_meshes_backward(fv, ours[0], gz, gb, gd, True, True)
b = mod.rasterize_meshes_backward(fv, ours[0], gz, gb, gd, True, True)
b2 = mod.rasterize_meshes_backward(fv, ours[0], gz, gb, gd, True, True)
v0, v1, v2 = fv[:, 0, :2], fv[:, 1, :2], fv[:, 2, :2]
area = ((v2[:, 0] - v0[:, 0]) * (v1[:, 1] - v0[:, 1]) - (v2[:, 1] - v0[:, 1]) * (v1[:, 0] - v0[:, 0])).abs()
e = torch.stack([(v1 - v0).norm(dim=1), (v2 - v0).norm(dim=1), (v2 - v1).norm(dim=1)], 1)
minedge = e.min(1).values
per = b.abs().amax(dim=(1, 2))
med = float(per[per > 0].median())
dev = (a - b).abs().amax(dim=(1, 2)) / (per + 1e-6 * med)
self_dev_ref = (b - b2).abs().amax(dim=(1, 2)) / (per + 1e-6 * med)
self_dev_ours = (a - a2).abs().amax(dim=(1, 2)) / (per + 1e-6 * med)
bad = dev > 5e-3
print("faces", F, "bad faces", int(bad.sum()), "nonfinite ours/ref", int((~torch.isfinite(a)).sum()), int((~torch.isfinite(b)).sum()))
print("ref run-to-run: faces beyond 5e-3:", int((self_dev_ref > 5e-3).sum()), "max", float(self_dev_ref.max()))
print("ours run-to-run: faces beyond 5e-3:", int((self_dev_ours > 5e-3).sum()), "max", float(self_dev_ours.max()))
sing = U.faces_with_singular_perspective(fv, ours[0])
print("flagged singular", int(sing.sum()), "bad & ~sing", int((bad & ~sing).sum()), "bad & sing", int((bad & sing).sum()))
rest = bad & ~sing
if rest.any():
    order = torch.argsort(dev * rest, descending=True)[:5]
    for i in order.tolist():
        print(" face", i, "dev", float(dev[i]), "area", float(area[i]), "ours", a[i].flatten()[:3].tolist(), "ref", b[i].flatten()[:3].tolist())
print("dev quantiles over ~sing:", torch.quantile(dev[~sing], torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device=d)).tolist())
import numpy as np
order = torch.argsort(dev * rest, descending=True)[:8]
dump = {}
for t, i in enumerate(order.tolist()):
    mask = ours[0] == i
    idx = mask.nonzero()
    dump[f"face{t}"] = np.array([i])
    dump[f"fv{t}"] = fv[i].cpu().numpy()
    dump[f"idx{t}"] = idx.cpu().numpy()
    dump[f"gz{t}"] = gz[mask].cpu().numpy()
    dump[f"gd{t}"] = gd[mask].cpu().numpy()
    dump[f"gb{t}"] = gb[mask].cpu().numpy()
    dump[f"zbuf{t}"] = ours[1][mask].cpu().numpy()
    dump[f"bary{t}"] = ours[2][mask].cpu().numpy()
    dump[f"dist{t}"] = ours[3][mask].cpu().numpy()
    dump[f"ours{t}"] = a[i].cpu().numpy()
    dump[f"ref{t}"] = b[i].cpu().numpy()
os.makedirs(os.path.join(ROOT, "gpurun_out", "r03"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "r03", "bwd_worst.npz"), **dump)
print("dumped", [int(x) for x in order.tolist()])
