"""Golden fixture for atlas texture sampling and hard blending, generated FROM THE REFERENCE (build container only).

    python tests/golden/make_golden_atlas_hard.py   ->  tests/golden/atlas_hard_ref.npz

The reference's own TexturesAtlas.sample_textures (pytorch3d/renderer/mesh/textures.py:565-612) and hard_rgb_blend
(pytorch3d/renderer/blending.py:54-88) with torch autograd, on fragments rasterized by the reference's C++ CPU kernel:
  * "blur": a blur band with clipped barycentrics (the rasterizer's default when blur > 0; coordinates reach 0 and 1
    exactly: the clamp to R - 1),
  * "hard": blur 0 (every sample inside its face),
  * "wild": synthetic barycentrics in [-0.45, 1.3] (negative cells wrap like torch's negative indices); the reference
    raises IndexError on part of such samples (with UNclipped blur-band barycentrics TexturesAtlas cannot be sampled at
    all at these sizes), so each candidate is tried alone and only the ones it can index are kept.
Atlas resolutions R = 1, 4 and 5, C = 3 (and C = 5 for the sampling alone).
"""
import os
import sys
from collections import namedtuple

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import make_golden as mg
    import _util as U

    ref = mg.bind_reference()
    from pytorch3d.renderer.blending import BlendParams, hard_rgb_blend
    from pytorch3d.renderer.mesh.textures import TexturesAtlas
    from pytorch3d.structures import Meshes

    gen = torch.Generator().manual_seed(4242)
    verts_l, faces_l = U.hetero_batch(2, seed=9, fmin=40, fmax=90)
    real = Meshes(verts=verts_l, faces=faces_l)
    fv = real.verts_packed()[real.faces_packed()]
    nf = [f.shape[0] for f in faces_l]
    N, H, W, K = 2, 17, 15, 3
    Frag = namedtuple("Frag", "pix_to_face bary_coords")
    out = {"num_faces": torch.tensor(nf)}
    for tag, blur in (("blur", 3e-3), ("hard", 0.0)):
        p2f, zbuf, bary, dists = ref.rasterize_meshes(fv, real.mesh_to_faces_packed_first_idx(), real.num_faces_per_mesh(),
                                                      torch.full((fv.shape[0],), -1, dtype=torch.int64), (H, W), blur, K,
                                                      0, 0, True, blur > 0, False)
        out.update({f"{tag}_pix_to_face": p2f, f"{tag}_bary": bary})
        for R, C in ((1, 3), (5, 3), (4, 5)):
            atlas = [torch.rand(n, R, R, C, generator=gen).requires_grad_(True) for n in nf]
            tex = TexturesAtlas(atlas=atlas)
            texels = tex.sample_textures(Frag(p2f, bary))
            g = torch.randn(texels.shape, generator=gen)
            (texels * g).sum().backward()
            key = f"{tag}_R{R}_C{C}"
            out.update({f"{key}_atlas": torch.cat([a.detach() for a in atlas]), f"{key}_texels": texels,
                        f"{key}_grad_texels": g, f"{key}_grad_atlas": torch.cat([a.grad for a in atlas])})
            print(key, "coverage", float((p2f >= 0).float().mean()), "bary range", float(bary[p2f >= 0].min()),
                  float(bary[p2f >= 0].max()))
        # hard_rgb_blend on random colours of the same fragments
        colors = torch.rand(N, H, W, K, 3, generator=gen).requires_grad_(True)
        bg = (0.25, 0.5, 0.875)
        img = hard_rgb_blend(colors, Frag(p2f, bary), BlendParams(background_color=bg))
        g = torch.randn(img.shape, generator=gen)
        (img * g).sum().backward()
        out.update({f"{tag}_colors": colors, f"{tag}_background": torch.tensor(bg), f"{tag}_image": img,
                    f"{tag}_grad_image": g, f"{tag}_grad_colors": colors.grad})
    # "wild": one sample per candidate, kept when the reference can index it
    Fw, R, C = 6, 4, 3
    atlas0 = torch.rand(Fw, R, R, C, generator=gen)
    cand_b = torch.rand(1500, 2, generator=gen) * 1.75 - 0.45
    cand_b = torch.cat([cand_b, 1.0 - cand_b.sum(-1, keepdim=True)], dim=-1)
    cand_f = torch.randint(0, Fw, (1500,), generator=gen)
    tex0 = TexturesAtlas(atlas=[atlas0])
    keep = []
    for i in range(cand_b.shape[0]):
        try:
            tex0.sample_textures(Frag(cand_f[i].view(1, 1, 1, 1), cand_b[i].view(1, 1, 1, 1, 3)))
            keep.append(i)
        except IndexError:
            pass
    keep = torch.tensor(keep)
    print("wild: reference indexes", len(keep), "of", cand_b.shape[0], "candidates; negative coordinates in",
          int((cand_b[keep, :2] < -1.0 / R).any(-1).sum()))
    p2f_w, bary_w = cand_f[keep].view(1, 1, -1, 1), cand_b[keep].view(1, 1, -1, 1, 3)
    at = atlas0.clone().requires_grad_(True)
    texels = TexturesAtlas(atlas=[at]).sample_textures(Frag(p2f_w, bary_w))
    g = torch.randn(texels.shape, generator=gen)
    (texels * g).sum().backward()
    out.update({"wild_pix_to_face": p2f_w, "wild_bary": bary_w, "wild_atlas": atlas0, "wild_texels": texels,
                "wild_grad_texels": g, "wild_grad_atlas": at.grad, "wild_rejected_bary": cand_b[[i for i in range(1500) if i not in set(keep.tolist())]],})
    mg.save("atlas_hard_ref", **out)


if __name__ == "__main__":
    main()
