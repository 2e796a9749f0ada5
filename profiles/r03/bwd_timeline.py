"""Per-wave start / end times of mesh_backward_rows_kernel on the bench launch (temporary probe build, not the product)."""
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
(p2f, zb, bary, dist), cover = _C._rasterize_meshes_covered(fv, first, cnt, nbr, (512, 512), blur, 8, 32, 64238, True, True, False)
gen = torch.Generator().manual_seed(1)
gz, gb, gd = [torch.randn(t.shape, generator=gen).to(d) for t in (zb, bary, dist)]
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * (16384 * 4 * 4))()
for use_cover in (False, True):
    for _ in range(3):
        _C.rasterize_meshes_backward(fv, p2f, gz, gb, gd, True, True, _cover=cover if use_cover else None)
    lib.p3d_tl_read(out)
    a = np.frombuffer(out, dtype=np.uint64).reshape(-1, 4)
    st, en, act = a[:, 0].astype(np.float64), a[:, 1].astype(np.float64), a[:, 2] > 0
    hw = a[:, 3]
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
    hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
    cu = (hwid >> 8) & 0xf
    sh = (hwid >> 12) & 0x1
    se = (hwid >> 13) & 0x7
    simd = (hwid >> 4) & 0x3
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    dur = (en - st)[act]
    print(f"cover={use_cover}: working waves {int(act.sum())}, duration mean {dur.mean():.0f} p50 {np.percentile(dur,50):.0f} p90 {np.percentile(dur,90):.0f} max {dur.max():.0f} ticks")
    cus = np.unique(key)
    spans, infl_all, infl_work, starts = [], np.zeros(20), np.zeros(20), np.zeros(20)
    for c in cus:  # s_memtime is not one clock across the chip: every CU is put on its own time axis
        mc = key == c
        t0 = st[mc].min()
        span = en[mc].max() - t0
        spans.append(span)
        mid = (np.arange(20) + 0.5) / 20 * span
        ww = mc & act
        infl_all += [((st[mc] - t0 < m_) & (en[mc] - t0 > m_)).sum() for m_ in mid]
        infl_work += [((st[ww] - t0 < m_) & (en[ww] - t0 > m_)).sum() for m_ in mid]
        starts += np.histogram(st[mc] - t0, bins=np.linspace(0, span, 21))[0]
    spans = np.array(spans)
    print(f"  {len(cus)} CUs; span per CU: min {spans.min():.0f} mean {spans.mean():.0f} max {spans.max():.0f} ticks")
    print("  mean waves in flight per CU (24 slots) at 20 points of the CU's own span, all waves:", (infl_all / len(cus)).round(1).tolist())
    print("  ... working waves:", (infl_work / len(cus)).round(1).tolist())
    print("  wave starts per CU per 1/20 span:", (starts / len(cus)).round(1).tolist())
    ks = key * 4 + simd
    mx_inflight, mean_inflight = [], []
    for c in np.unique(ks)[:256]:
        mc = ks == c
        ev = np.concatenate([np.stack([st[mc], np.ones(mc.sum())], 1), np.stack([en[mc], -np.ones(mc.sum())], 1)])
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        run = np.cumsum(ev[:, 1])
        mx_inflight.append(run.max())
        mean_inflight.append(((en - st)[mc].sum()) / (en[mc].max() - st[mc].min()))
    print("  per SIMD (first 256 SIMDs): waves in flight max over time: min %d median %d max %d; time-mean %.1f" % (
        min(mx_inflight), np.median(mx_inflight), max(mx_inflight), np.mean(mean_inflight)))
    wgs = np.arange(len(st)) // 4
    same = [len(set(simd[i * 4:(i + 1) * 4])) for i in range(0, 4000, 7)]
    print("  distinct SIMDs among the 4 waves of a workgroup (sample):", np.bincount(same).tolist())
    per_cu = np.array([(en - st)[(key == c) & act].sum() for c in cus]) / spans.mean()
    print("  time-mean working waves in flight per CU: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % (
        per_cu.min(), np.percentile(per_cu, 10), np.median(per_cu), np.percentile(per_cu, 90), per_cu.max()))
    nall = np.array([(key == c).sum() for c in cus]); nwork = np.array([((key == c) & act).sum() for c in cus])
    order = np.argsort(per_cu)
    print("  CUs with the least working waves: (key xcc/se/sh/cu, waves, working)", [(int(cus[i]) >> 8, (int(cus[i]) >> 5) & 7, (int(cus[i]) >> 4) & 1, int(cus[i]) & 15, int(nall[i]), int(nwork[i])) for i in order[:12]])
    print("  CUs with the most: ", [(int(cus[i]) >> 8, (int(cus[i]) >> 5) & 7, (int(cus[i]) >> 4) & 1, int(cus[i]) & 15, int(nall[i]), int(nwork[i])) for i in order[-6:]])
    print("  waves per CU: min %d median %d max %d; distinct raw hwid CU fields: cu %s sh %s se %s" % (nall.min(), np.median(nall), nall.max(), np.unique(cu).tolist(), np.unique(sh).tolist(), np.unique(se).tolist()))
    print("  sum of working-wave durations / (mean span x CUs x 24 slots) = %.2f" % (dur.sum() / (spans.mean() * len(cus) * 24)))
