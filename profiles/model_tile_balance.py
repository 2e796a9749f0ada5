#!/usr/bin/env python
"""CPU model behind DESIGN.md 8.1 / 8.5: how evenly is the fine rasterizer's work spread inside a workgroup, and how large
are the union rectangles the binning's waves walk?  (No GPU: geometry of the bench batch only.)

    python profiles/model_tile_balance.py [--meshes 64]

(1) A workgroup of mesh_fine owns a 16x16 tile, its four waves the four 8x8 sub-tiles, and they share the staged list:
    the workgroup lasts as long as its busiest wave.  Counting, per sub-tile, the faces whose blur-expanded bounding box
    touches it (what a wave visits before depth culling), sum / (4 * max) over the active tiles is the share of the wave
    slots that do work: 0.85 on the bench batch -- at least 15 % idle at the chunk barriers.
(2) bin_count / bin_fill walk, per wave of 64 consecutive faces, the union rectangle of their bins with one ballot per
    bin: 67 bins on average against 51 distinct bins touched (401 (face, bin) members): the exhaustive walk is not the waste.
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
    ap.add_argument("--meshes", type=int, default=64)
    ap.add_argument("--image-size", type=int, default=512)
    args = ap.parse_args()
    import _util as U

    verts, faces = U.hetero_batch(args.meshes, seed=0)
    blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
    r = math.sqrt(blur)
    W = args.image_size
    sub, tot_max, tot_sum, tiles = W // 8, 0, 0, 0
    n_waves, tot_union, tot_distinct, tot_members = 0, 0, 0, 0
    for n in range(args.meshes):
        fv = verts[n].numpy()[faces[n].numpy()]
        x, y = fv[:, :, 0], fv[:, :, 1]
        xlo, xhi, ylo, yhi = x.min(1) - r, x.max(1) + r, y.min(1) - r, y.max(1) + r
        ok = fv[:, :, 2].min(1) > 1e-8
        # pixel centres: ndc = -1 + (2 i + 1) / W
        lo = lambda a: np.clip(np.ceil(((a + 1) * W - 1) / 2), 0, W - 1).astype(int)  # noqa: E731
        hi = lambda a: np.clip(np.floor(((a + 1) * W - 1) / 2), 0, W - 1).astype(int)  # noqa: E731
        ix0, ix1, iy0, iy1 = lo(xlo), hi(xhi), lo(ylo), hi(yhi)
        ok &= (ix1 >= ix0) & (iy1 >= iy0)
        cnt = np.zeros((sub, sub), dtype=np.int64)
        for a, b, c, d in zip(ix0[ok] // 8, ix1[ok] // 8, iy0[ok] // 8, iy1[ok] // 8):
            cnt[c:d + 1, a:b + 1] += 1
        t = cnt.reshape(sub // 2, 2, sub // 2, 2).transpose(0, 2, 1, 3).reshape(sub // 2, sub // 2, 4)
        mx, sm = t.max(-1), t.sum(-1)
        act = mx > 0
        tot_max += int((4 * mx[act]).sum())
        tot_sum += int(sm[act].sum())
        tiles += int(act.sum())
        # binning: 16-pixel bins, waves of 64 consecutive faces inside chunks of 1024
        bx0, bx1, by0, by1 = ix0 // 16, ix1 // 16, iy0 // 16, iy1 // 16
        F = fv.shape[0]
        for c0 in range(0, F, 1024):
            for w0 in range(c0, min(c0 + 1024, F), 64):
                s = slice(w0, min(w0 + 64, F))
                m = ok[s]
                if not m.any():
                    continue
                a0, a1, b0, b1 = bx0[s][m], bx1[s][m], by0[s][m], by1[s][m]
                tot_union += int((a1.max() - a0.min() + 1) * (b1.max() - b0.min() + 1))
                seen = set()
                for p, q, u, v in zip(a0, a1, b0, b1):
                    tot_members += int((q - p + 1) * (v - u + 1))
                    for yy in range(u, v + 1):
                        seen.update(range(yy * 64 + p, yy * 64 + q + 1))
                tot_distinct += len(seen)
                n_waves += 1
    print(f"mesh_fine: {tiles} active 16x16 tiles; candidate faces per sub-tile wave: sum {tot_sum}, 4 * max {tot_max} "
          f"-> share of the wave slots with work {tot_sum / tot_max:.3f}")
    print(f"binning: {n_waves} waves of 64 faces; bins walked (union rectangle) {tot_union / n_waves:.1f} per wave, distinct bins "
          f"touched {tot_distinct / n_waves:.1f}, (face, bin) members {tot_members / n_waves:.1f}")


if __name__ == "__main__":
    main()
