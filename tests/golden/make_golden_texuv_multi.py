"""Golden fixture for multi-map UV sampling (TexturesUV with maps_ids), generated FROM THE REFERENCE (build container only).

    python tests/golden/make_golden_texuv_multi.py   ->  tests/golden/texuv_multi_ref.npz

The reference's own TexturesUV.sample_textures (pytorch3d/renderer/mesh/textures.py:1270-1313: the map index of each
face becomes the z coordinate of a 3-D F.grid_sample over (N, C, M, Hm, Wm)) with torch autograd, on fragments
rasterized by the reference's C++ CPU kernel.  Two meshes with the SAME number of faces (the reference indexes the
flattened padded maps_ids with packed face indices, which only agrees with the padded layout then), M = 3 maps.
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

CASES = {"default": dict(padding_mode="border", align_corners=True, sampling_mode="bilinear"),
         "zeros_noalign": dict(padding_mode="zeros", align_corners=False, sampling_mode="bilinear"),
         "border_noalign": dict(padding_mode="border", align_corners=False, sampling_mode="bilinear"),
         "nearest": dict(padding_mode="zeros", align_corners=True, sampling_mode="nearest")}


def main():
    import make_golden as mg
    import _util as U

    ref = mg.bind_reference()
    from pytorch3d.renderer.mesh.textures import TexturesUV
    from pytorch3d.structures import Meshes

    gen = torch.Generator().manual_seed(909)
    v, f = U.ico_sphere(1)  # 80 faces
    verts_l = [U.to_ndc(v), U.to_ndc(v * 0.7 + torch.tensor([0.2, -0.1, 0.0]))]
    faces_l = [f, f]
    real = Meshes(verts=verts_l, faces=faces_l)
    fv = real.verts_packed()[real.faces_packed()]
    N, H, W, K, M = 2, 19, 23, 2, 3
    nf = [x.shape[0] for x in faces_l]
    p2f, zbuf, bary, dists = ref.rasterize_meshes(fv, real.mesh_to_faces_packed_first_idx(), real.num_faces_per_mesh(),
                                                  torch.full((fv.shape[0],), -1, dtype=torch.int64), (H, W), 2e-4, K, 0, 0,
                                                  True, False, False)
    Frag = namedtuple("Frag", "pix_to_face bary_coords")
    verts_uvs0 = [torch.rand(3 * n, 2, generator=gen) * 1.3 - 0.15 for n in nf]
    faces_uvs = [torch.arange(3 * n).view(n, 3) for n in nf]
    maps0 = torch.rand(N, M, 6, 9, 3, generator=gen)
    maps_ids = torch.randint(0, M, (N, nf[0]), generator=gen)
    out = {"pix_to_face": p2f, "bary": bary, "faces_uvs_0": faces_uvs[0], "faces_uvs_1": faces_uvs[1],
           "verts_uvs": torch.cat(verts_uvs0), "maps": maps0, "maps_ids": maps_ids, "num_faces": torch.tensor(nf)}
    for tag, cfg in CASES.items():
        vu = [x.clone().requires_grad_(True) for x in verts_uvs0]
        mp = maps0.clone().requires_grad_(True)
        b = bary.clone().requires_grad_(True)
        tex = TexturesUV(maps=mp, faces_uvs=faces_uvs, verts_uvs=vu, maps_ids=maps_ids, **cfg)
        texels = tex.sample_textures(Frag(p2f, b))
        g = torch.randn(texels.shape, generator=gen)
        (texels * g).sum().backward()
        out.update({f"{tag}_texels": texels, f"{tag}_grad_texels": g, f"{tag}_grad_maps": mp.grad,
                    f"{tag}_grad_verts_uvs": torch.cat([x.grad if x.grad is not None else torch.zeros_like(x) for x in vu]),
                    f"{tag}_grad_bary": b.grad if b.grad is not None else torch.zeros_like(b)})
    mg.save("texuv_multi_ref", **out)
    print("coverage", float((p2f >= 0).float().mean()))


if __name__ == "__main__":
    main()
