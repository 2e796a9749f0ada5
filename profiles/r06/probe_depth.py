"""Wave-level counters of a depth pre-test in the fine kernel's candidate loop (temporary -DP3D_PROBE build of raster_mesh.hip,
profiles/r06/probe_depth.patch; not the product).  Question: of the evaluations in which no lane can enter its queue after the
depth, how many could a cheap bound (P (1 - m) > kth_z Q, all operands >= 0) plus the exact vertex-region rule (one positive
barycentric: depth = that vertex's depth, bit for bit) reject for EVERY lane ahead of the two reciprocal chains?"""
import ctypes, sys, os, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util as U, pytorch3d_amd as p3d
from pytorch3d_amd import _C, _lib
d = torch.device("cuda:0")
verts, faces = U.hetero_batch(64, seed=0, torus_div=1.0)
m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
fv = m.verts_packed()[m.faces_packed()].contiguous()
first, cnt = m.mesh_to_faces_packed_first_idx(), m.num_faces_per_mesh()
nbr = torch.full((fv.shape[0],), -1, dtype=torch.int64, device=d)
blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 16)()
M = int(max(10000, fv.shape[0] / 5))
_C.rasterize_meshes(fv, first, cnt, nbr, (512, 512), blur, 8, 32, M, True, True, False)
lib.p3d_probe_read(out)
_C.rasterize_meshes(fv, first, cnt, nbr, (512, 512), blur, 8, 32, M, True, True, False)
lib.p3d_probe_read(out)
v = list(out)
names = ["evaluations (wave level)", "  no lane admitted after the exact depth", "  every lane rejected by the bound, margin 2e-5",
         "  every lane rejected by the bound, margin 2e-6", "  every lane rejected by bound 2e-6 or vertex rule",
         "VIOLATIONS (lanes rejected that the exact depth admits)", "active lanes", "lanes rejected by bound 2e-6 or vertex rule",
         "all-fail evaluations the rule leaves standing", "  their undecided lanes: one positive barycentric",
         "  two positive", "  three positive (inside)", "  undecided lanes that tie exactly with the K-th depth", "  undecided lanes in total"]
for n, x in zip(names, v):
    print(f"{n:<70}{x}")
h = (ctypes.c_ulonglong * 66)()
lib.p3d_probe_hist(h)
h = list(h)
print("inserting passes", h[64], "inserting lanes", h[65], "=> %.1f lanes / inserting pass" % (h[65] / max(h[64], 1)))
print("rows: lowest insertion position over the wave's inserting lanes (pmin); columns: highest entry any inserting lane has filled, capped at 7 (cmax)")
tot = sum(h[:64])
steps_now, steps_new = 0, 0
for r in range(8):
    print("pmin %d: " % r + " ".join("%8d" % h[r * 8 + c] for c in range(8)))
    for c in range(8):
        steps_now += 8 * h[r * 8 + c]
        steps_new += (max(c, r) - r + 1) * h[r * 8 + c]
print("network steps: today %d, with start at cmax and exit at pmin %d (%.3f)" % (steps_now, steps_new, steps_new / max(steps_now, 1)))
pm = [sum(h[r * 8:(r + 1) * 8]) for r in range(8)]
print("pmin histogram", pm, "exit-at-pmin only: %.3f" % (sum((8 - r) * pm[r] for r in range(8)) / max(8 * tot, 1)))
