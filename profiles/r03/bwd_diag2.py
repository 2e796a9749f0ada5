import math, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util as U
from oracle import oracle as orc
from oracle.backward_f64 import backward_f64
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
b = mod.rasterize_meshes_backward(fv, ours[0], gz, gb, gd, True, True)
truth, abs_sum = backward_f64(fv, ours[0], gz, gb, gd, True, True)
nz = abs_sum[abs_sum > 0]
scale = abs_sum + 1e-7 * float(nz.median())
dev = (a.double() - truth).abs() / scale
sing = U.faces_with_singular_perspective(fv, ours[0], rel=1e-4)
badf = (dev > 5e-3).reshape(F, -1).any(1)
print("bad faces", int(badf.sum()), "of them flagged singular", int((badf & sing).sum()), "flagged total", int(sing.sum()))
for i in badf.nonzero().flatten().tolist()[:12]:
    j = int(dev[i].flatten().argmax())
    print(" face", i, "entry", j, "dev", float(dev[i].flatten()[j]), "truth", float(truth[i].flatten()[j]), "ours", float(a[i].flatten()[j]),
          "ref", float(b[i].flatten()[j]), "abs_sum", float(abs_sum[i].flatten()[j]), "samples", int((ours[0] == i).sum()), "sing", bool(sing[i]))
