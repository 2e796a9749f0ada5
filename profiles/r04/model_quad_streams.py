#!/usr/bin/env python
"""CPU estimate: evaluation passes of the fine rasterizer if the four 4x4 quads of a wave's 8x8 sub-tile walked their OWN
candidate lists (a lane evaluates its quad's current candidate) instead of all 64 lanes visiting every candidate of the
sub-tile.  Geometry of the bench batch only (blur-expanded bounding boxes; no depth culling, no triangle-level pruning), as
profiles/model_mask_pairs.py.   python profiles/r04/model_quad_streams.py [--meshes 8] [--torus-div 1.0]"""
import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--meshes", type=int, default=8)
    ap.add_argument("--torus-div", type=float, default=1.0)
    args = ap.parse_args()
    import _util as U

    verts, faces = U.hetero_batch(args.meshes, seed=0, torus_div=args.torus_div)
    r = math.sqrt(math.log(1.0 / 1e-4 - 1.0) * 1e-4)
    W = 512
    cands = lanes = 0
    passes = {"quads4x4": 0, "halves8x4": 0, "halves4x8": 0, "pairs": 0}
    lanes_q = 0
    for n in range(args.meshes):
        fv = verts[n].numpy()[faces[n].numpy()]
        x, y = fv[:, :, 0], fv[:, :, 1]
        lo = lambda a: np.clip(np.ceil(((a + 1) * W - 1) / 2), 0, W - 1).astype(int)  # noqa: E731
        hi = lambda a: np.clip(np.floor(((a + 1) * W - 1) / 2), 0, W - 1).astype(int)  # noqa: E731
        ix0, ix1, iy0, iy1 = lo(x.min(1) - r), hi(x.max(1) + r), lo(y.min(1) - r), hi(y.max(1) + r)
        per = {}
        for f in np.nonzero((ix1 >= ix0) & (iy1 >= iy0))[0]:
            a0, a1, b0, b1 = int(ix0[f]), int(ix1[f]), int(iy0[f]), int(iy1[f])
            for sy in range(b0 // 8, b1 // 8 + 1):
                for sx in range(a0 // 8, a1 // 8 + 1):
                    c0, c1 = max(a0, sx * 8) - sx * 8, min(a1, sx * 8 + 7) - sx * 8
                    r0, r1 = max(b0, sy * 8) - sy * 8, min(b1, sy * 8 + 7) - sy * 8
                    per.setdefault((sy, sx), []).append((c0, c1, r0, r1))
        for ms in per.values():
            q = [0, 0, 0, 0]
            hx = [0, 0]
            hy = [0, 0]
            masks = []
            for c0, c1, r0, r1 in ms:
                lanes += (c1 - c0 + 1) * (r1 - r0 + 1)
                tx = [c0 <= 3, c1 >= 4]
                ty = [r0 <= 3, r1 >= 4]
                for qy in range(2):
                    for qx in range(2):
                        if tx[qx] and ty[qy]:
                            q[qy * 2 + qx] += 1
                            cc0, cc1 = max(c0, 4 * qx), min(c1, 4 * qx + 3)
                            rr0, rr1 = max(r0, 4 * qy), min(r1, 4 * qy + 3)
                            lanes_q += (cc1 - cc0 + 1) * (rr1 - rr0 + 1)
                for h in range(2):
                    hx[h] += tx[h]
                    hy[h] += ty[h]
                cm = ((1 << (c1 + 1)) - 1) & ~((1 << c0) - 1)
                m = 0
                for rr in range(r0, r1 + 1):
                    m |= cm << (8 * rr)
                masks.append(m)
            passes["quads4x4"] += max(q)
            passes["halves4x8"] += max(hx)
            passes["halves8x4"] += max(hy)
            used = [False] * len(masks)
            p = 0
            for i, m in enumerate(masks):
                if used[i]:
                    continue
                p += 1
                for j in range(i + 1, len(masks)):
                    if not used[j] and (masks[j] & m) == 0:
                        used[j] = True
                        break
            passes["pairs"] += p
            cands += len(ms)
    print(f"torus_div {args.torus_div}, {args.meshes} meshes: {cands} (face, sub-tile) candidates, {lanes / cands:.1f} of 64 lanes inside the box")
    for k, v in passes.items():
        print(f"  {k:10s}: {v} passes = {v / cands:.3f} of the candidates" + (f"; lanes in box per pass {lanes_q / v:.1f}" if k == "quads4x4" else ""))


if __name__ == "__main__":
    main()
