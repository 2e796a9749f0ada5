"""mesh_backward_rows_kernel with parts switched off (temporary -DP3D_ABLATION build with the P3D_DEBUG_BWD bits wired into
the rows kernel; results are wrong by construction, only the times mean something)."""
import sys, os, math, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _util as U, pytorch3d_amd as p3d
    from pytorch3d_amd import _C
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
    for _ in range(3):
        _C.rasterize_meshes_backward(fv, p2f, gz, gb, gd, True, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        _C.rasterize_meshes_backward(fv, p2f, gz, gb, gd, True, True)
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"bits": int(os.environ.get("P3D_DEBUG_BWD", "0")), "ms": e0.elapsed_time(e1) / 20}))
else:
    names = {0: "full", 2: "no table", 3: "no table, no arithmetic (all loads)", 7: "... and no face gather (2 rows only)",
             11: "no table/arith, no gradient streams (p2f + gather)", 15: "pix_to_face stream only", 1: "no arithmetic (loads + table)",
             4: "full but gather from 2 rows", 16: "table without the list summation", 32: "table without the slot update",
             48: "table: probe + exchange only", 64: "no global atomics in the flush", 17: "no arithmetic, no list summation",
             49: "no arithmetic; probe + exchange only"}
    for bits, name in names.items():
        env = dict(os.environ, P3D_DEBUG_BWD=str(bits))
        out = subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
        print(f"{bits:3d} {name:55s} {out}", flush=True)
