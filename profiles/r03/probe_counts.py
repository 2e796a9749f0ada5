"""Wave-level counters of the fine kernel's candidate loop (temporary -DPROBE build, not the product): how many candidate
visits reach the evaluation, how many evaluations end with a hit / an insertion in at least one lane."""
import ctypes, sys, os, math, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util as U, pytorch3d_amd as p3d
from pytorch3d_amd import _C, _lib
d = torch.device("cuda:0")
verts, faces = U.hetero_batch(64, seed=0)
m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
fv = m.verts_packed()[m.faces_packed()].contiguous()
first, cnt = m.mesh_to_faces_packed_first_idx(), m.num_faces_per_mesh()
nbr = torch.full((fv.shape[0],), -1, dtype=torch.int64, device=d)
blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
_C.rasterize_meshes(fv, first, cnt, nbr, (512, 512), blur, 8, 32, 64238, True, True, False)
lib.p3d_probe_read(out)
_C.rasterize_meshes(fv, first, cnt, nbr, (512, 512), blur, 8, 32, 64238, True, True, False)
lib.p3d_probe_read(out)
v = list(out)
print("candidate visits (wave level)", v[0])
print("  evaluated (some lane in box and not too deep)", v[1], "lanes active in those", v[5], "=> %.1f lanes / eval" % (v[5] / max(v[1], 1)))
print("  some lane could admit after pz (z >= 0, sorts before K-th)", v[2])
print("  some lane hit", v[3])
print("  some lane inserted", v[4], "lanes inserting", v[6], "=> %.1f lanes / inserting eval" % (v[6] / max(v[4], 1)))
