import sys, os, math, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["P3D_FWD_PROBE_OUT"] = "/tmp/probe.bin"
import _util as U, pytorch3d_amd as p3d
from pytorch3d_amd import _C
d = torch.device("cuda:0")
verts, faces = U.hetero_batch(64, seed=0)
m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
fv = m.verts_packed()[m.faces_packed()].contiguous()
first, cnt = m.mesh_to_faces_packed_first_idx(), m.num_faces_per_mesh()
nbr = torch.full((fv.shape[0],), -1, dtype=torch.int64, device=d)
blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
for _ in range(3):
    _C.rasterize_meshes(fv, first, cnt, nbr, (512, 512), blur, 8, 32, 64238, True, True, False)
torch.cuda.synchronize()
a = np.fromfile("/tmp/probe.bin", dtype=np.uint64).reshape(-1, 8).astype(np.float64)
act = a[a[:, 0] > 0]
tot, w, stage, count = act[:, 0], act[:, 1:5], act[:, 5], act[:, 6]
print("active workgroups", len(act), "mean total ticks", tot.mean(), "sum total", tot.sum())
print("share of workgroup time: staging+order %.3f, wave_chunk max-wave %.3f, mean-wave %.3f, rest %.3f" % (
    stage.sum() / tot.sum(), w.max(1).sum() / tot.sum(), w.mean(1).sum() / tot.sum(), 1 - (stage.sum() + w.max(1).sum()) / tot.sum()))
print("imbalance: sum(max wave) / sum(mean wave) = %.3f" % (w.max(1).sum() / w.mean(1).sum()))
for lo, hi in ((1, 64), (64, 128), (128, 256), (256, 384), (384, 10000)):
    s = (count >= lo) & (count < hi)
    if s.any():
        print(f"faces [{lo},{hi}): {int(s.sum())} wgs, mean total {tot[s].mean():.0f} ticks, share of all time {tot[s].sum() / tot.sum():.3f}, max/mean wave {w[s].max(1).sum() / max(w[s].mean(1).sum(), 1):.2f}, staging share {stage[s].sum() / tot[s].sum():.2f}")
