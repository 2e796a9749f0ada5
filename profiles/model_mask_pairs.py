#!/usr/bin/env python
"""CPU estimate behind profiles/next/pixel_pairs.patch: how many candidate evaluations of the fine rasterizer could share a
pass with another candidate whose pixel mask does not intersect theirs?  (Geometry of the bench batch only: the 64-bit
masks of the blur-expanded bounding boxes over every 8x8 sub-tile they touch, greedy first-fit pairing in list order; no
depth culling, no triangle-level pruning.)

    python profiles/model_mask_pairs.py [--meshes 8]
"""
import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--meshes", type=int, default=8)
    args = ap.parse_args()
    import _util as U

    verts, faces = U.hetero_batch(args.meshes, seed=0)
    r = math.sqrt(math.log(1.0 / 1e-4 - 1.0) * 1e-4)
    W = 512
    cands = pairs = lanes = full = 0
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
                    cm = ((1 << (c1 + 1)) - 1) & ~((1 << c0) - 1)
                    m = 0
                    for rr in range(r0, r1 + 1):
                        m |= cm << (8 * rr)
                    per.setdefault((sy, sx), []).append(m)
        for ms in per.values():
            used = [False] * len(ms)
            for i, m in enumerate(ms):
                pc = bin(m).count("1")
                lanes += pc
                full += pc == 64
                if used[i]:
                    continue
                for j in range(i + 1, len(ms)):
                    if not used[j] and (ms[j] & m) == 0:
                        used[i] = used[j] = True
                        pairs += 1
                        break
            cands += len(ms)
    print(f"{cands} (face, sub-tile) candidates on {args.meshes} bench meshes: {lanes / cands:.1f} of 64 pixels inside the box on "
          f"average, {full / cands:.2f} cover the whole sub-tile; greedy disjoint pairs {pairs} -> {pairs / cands:.3f} of the "
          f"evaluation passes saved")


if __name__ == "__main__":
    main()
