"""Per-wave time split of mesh_backward_rows_kernel (temporary probe build, not the product): s_memtime around the
operand loads + per-sample arithmetic and around the table step."""
import ctypes, sys, os, math, torch
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
p2f, zb, bary, dist = _C.rasterize_meshes(fv, first, cnt, nbr, (512, 512), blur, 8, 32, 64238, True, True, False)
gen = torch.Generator().manual_seed(1)
gz, gb, gd = [torch.randn(t.shape, generator=gen).to(d) for t in (zb, bary, dist)]
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
for _ in range(2):
    _C.rasterize_meshes_backward(fv, p2f, gz, gb, gd, True, True)
    lib.p3d_bprobe_read(out)
v = list(out)
print("waves with work", v[7], "steps", v[4], "steps with a sample", v[5], "samples", v[6], "=> %.1f lanes / step" % (v[6] / max(v[5], 1)))
print("ticks: total %d, loads+arithmetic %d (%.2f), table add %d (%.2f), final flush %d (%.2f), rest %.2f" % (
    v[0], v[1], v[1] / v[0], v[2], v[2] / v[0], v[3], v[3] / v[0], 1 - (v[1] + v[2] + v[3]) / v[0]))
print("per step with a sample: loads+arithmetic %.0f ticks, table add %.0f ticks" % (v[1] / v[5], v[2] / v[5]))
