"""Golden fixture for UV texture sampling, generated FROM THE REFERENCE (build container only).

    python tests/golden/make_golden_texuv.py   ->  tests/golden/texuv_ref.npz

The reference's own TexturesUV.sample_textures (pytorch3d/renderer/mesh/textures.py:1190-1268: interpolate + lerp +
F.grid_sample) with torch autograd, on fragments rasterized by the reference's C++ CPU kernel, for the default
configuration (bilinear, border, align_corners=True) and the other padding / alignment / sampling combinations.
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
    from pytorch3d.renderer.mesh.textures import TexturesUV
    from pytorch3d.structures import Meshes

    gen = torch.Generator().manual_seed(808)
    verts_l, faces_l = U.hetero_batch(2, seed=6, fmin=120, fmax=300)
    real = Meshes(verts=verts_l, faces=faces_l)
    fv = real.verts_packed()[real.faces_packed()]
    N, H, W, K = 2, 22, 26, 3
    p2f, zbuf, bary, dists = ref.rasterize_meshes(fv, real.mesh_to_faces_packed_first_idx(), real.num_faces_per_mesh(),
                                                  torch.full((fv.shape[0],), -1, dtype=torch.int64), (H, W), 2e-4, K, 0, 0,
                                                  True, False, False)  # unclipped barycentrics: uvs leave [0, 1] in the blur band
    Frag = namedtuple("Frag", "pix_to_face bary_coords")
    # a uv per face corner (every corner its own uv vertex), some outside [0, 1] to reach the padding
    nf = [f.shape[0] for f in faces_l]
    verts_uvs0 = [torch.rand(3 * n, 2, generator=gen) * 1.3 - 0.15 for n in nf]
    faces_uvs = [torch.arange(3 * n).view(n, 3) for n in nf]
    maps0 = torch.rand(N, 9, 13, 3, generator=gen)
    out = {"pix_to_face": p2f, "bary": bary, "faces_uvs_0": faces_uvs[0], "faces_uvs_1": faces_uvs[1],
           "verts_uvs": torch.cat(verts_uvs0), "maps": maps0, "num_faces": torch.tensor(nf)}
    cases = {"default": dict(padding_mode="border", align_corners=True, sampling_mode="bilinear"),
             "zeros_noalign": dict(padding_mode="zeros", align_corners=False, sampling_mode="bilinear"),
             "border_noalign": dict(padding_mode="border", align_corners=False, sampling_mode="bilinear"),
             "nearest": dict(padding_mode="zeros", align_corners=True, sampling_mode="nearest")}
    for tag, cfg in cases.items():
        vu = [x.clone().requires_grad_(True) for x in verts_uvs0]
        mp = maps0.clone().requires_grad_(True)
        b = bary.clone().requires_grad_(True)
        tex = TexturesUV(maps=mp, faces_uvs=faces_uvs, verts_uvs=vu, **cfg)
        tex._num_faces_per_mesh = nf
        texels = tex.sample_textures(Frag(p2f, b))
        g = torch.randn(texels.shape, generator=gen)
        (texels * g).sum().backward()
        out.update({f"{tag}_texels": texels, f"{tag}_grad_texels": g, f"{tag}_grad_maps": mp.grad,
                    f"{tag}_grad_verts_uvs": torch.cat([x.grad if x.grad is not None else torch.zeros_like(x) for x in vu]),
                    f"{tag}_grad_bary": b.grad if b.grad is not None else torch.zeros_like(b)})
    mg.save("texuv_ref", **out)
    print("coverage", float((p2f >= 0).float().mean()))


if __name__ == "__main__":
    main()
