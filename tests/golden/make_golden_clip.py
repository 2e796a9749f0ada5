"""Golden fixture for the clipping row, generated FROM THE REFERENCE (build container only).

    python tests/golden/make_golden_clip.py   ->  tests/golden/clip_ref.npz

The reference's clip_faces / convert_clipped_rasterization_to_original_faces (pytorch3d/renderer/mesh/clip.py) and
its whole rasterize_meshes(z_clip_value=..., cull_to_frustum=...) pipeline (rasterize_meshes.py:144-250) on the
reference's CPU kernels, with torch-autograd gradients back to the vertices.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def scene(gen, F, zlo, zhi):
    """Triangles around the clipping plane: xy in NDC-ish range, z straddling 0.3."""
    c = torch.rand(F, 1, 2, generator=gen) * 2.0 - 1.0
    xy = c + (torch.rand(F, 3, 2, generator=gen) - 0.5) * 0.9
    z = torch.rand(F, 3, 1, generator=gen) * (zhi - zlo) + zlo
    return torch.cat([xy, z], -1).float()


def main():
    import make_golden as mg

    mg.bind_reference()
    from pytorch3d.renderer.mesh.clip import ClipFrustum, clip_faces, convert_clipped_rasterization_to_original_faces
    from pytorch3d.renderer.mesh.rasterize_meshes import rasterize_meshes
    from pytorch3d.structures import Meshes

    gen = torch.Generator().manual_seed(99)
    out = {}
    # ---- clip_faces alone: every field of ClippedFaces, gradients of a random functional of its outputs ----------
    cases = {"a": dict(persp=True, cull=True, z_clip=0.3), "b": dict(persp=False, cull=False, z_clip=0.3),
             "c": dict(persp=True, cull=True, z_clip=None), "d": dict(persp=False, cull=True, z_clip=0.5)}
    for tag, cfg in cases.items():
        F = 60
        fv = scene(gen, F, -0.4, 1.6).requires_grad_(True)
        first = torch.tensor([0, 25, 25])  # the middle mesh is empty
        count = torch.tensor([25, 0, 35])
        fr = ClipFrustum(left=-1, right=1, top=-1, bottom=1, znear=None, zfar=None, perspective_correct=cfg["persp"],
                         cull=cfg["cull"], z_clip_value=cfg["z_clip"])
        cf = clip_faces(fv, first, count, frustum=fr)
        g_fv = torch.randn(cf.face_verts.shape, generator=gen)
        loss = (cf.face_verts * g_fv).sum()
        g_bc = None
        if cf.barycentric_conversion is not None:
            g_bc = torch.randn(cf.barycentric_conversion.shape, generator=gen)
            loss = loss + (cf.barycentric_conversion * g_bc).sum()
        loss.backward()
        rec = dict(face_verts=fv, first=first, count=count, persp=cfg["persp"], cull=cfg["cull"],
                   z_clip=-1e30 if cfg["z_clip"] is None else cfg["z_clip"], has_z_clip=cfg["z_clip"] is not None,
                   out_face_verts=cf.face_verts, out_first=cf.mesh_to_face_first_idx, out_count=cf.num_faces_per_mesh,
                   grad_out_face_verts=g_fv, grad_face_verts=fv.grad)
        for name in ("faces_clipped_to_unclipped_idx", "barycentric_conversion", "faces_clipped_to_conversion_idx",
                     "clipped_faces_neighbor_idx"):
            v = getattr(cf, name)
            rec["has_" + name] = v is not None
            if v is not None:
                rec[name] = v
        if g_bc is not None:
            rec["grad_barycentric_conversion"] = g_bc
        # convert: a synthetic "rasterization" that references every clipped face
        Fc = cf.face_verts.shape[0]
        p2f = torch.randint(-1, Fc, (2, 6, 5, 3), generator=gen)
        bary = torch.rand(2, 6, 5, 3, 3, generator=gen)
        p2f_u, bary_u = convert_clipped_rasterization_to_original_faces(p2f, bary, cf)
        rec.update(conv_p2f=p2f, conv_bary=bary, conv_out_p2f=p2f_u, conv_out_bary=bary_u.detach())
        out.update({f"{tag}_{k}": v for k, v in rec.items()})

    # ---- the whole pipeline: rasterize_meshes with clipping, CPU kernels, gradients to the vertices -----------------
    # (perspective_correct and clip_barycentric_coords are never both on: the reference's CPU backward clips on the
    # corrected barycentrics, its CUDA backward -- which the HIP kernels follow -- on the uncorrected ones,
    # rasterize_meshes_cpu.cpp:499 vs rasterize_meshes.cu:528)
    for tag, cfg in {"p": dict(persp=True, blur=1e-3, K=4, clipb=False, cull=True),
                     "q": dict(persp=False, blur=2e-3, K=3, clipb=True, cull=False)}.items():
        V = 40
        verts = torch.cat([torch.rand(V, 2, generator=gen) * 2.2 - 1.1, torch.rand(V, 1, generator=gen) * 2.0 - 0.3], 1)
        verts = verts.requires_grad_(True)
        faces = torch.randint(0, V, (50, 3), generator=gen)
        faces = faces[(faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])]
        meshes = Meshes(verts=[verts], faces=[faces])
        p2f, zbuf, bary, dists = rasterize_meshes(meshes, image_size=(24, 20), blur_radius=cfg["blur"],
                                                  faces_per_pixel=cfg["K"], bin_size=0,
                                                  perspective_correct=cfg["persp"],
                                                  clip_barycentric_coords=cfg["clipb"], z_clip_value=0.25,
                                                  cull_to_frustum=cfg["cull"])
        gz = torch.randn(zbuf.shape, generator=gen)
        gb = torch.randn(bary.shape, generator=gen)
        gd = torch.randn(dists.shape, generator=gen)
        ((zbuf * gz).sum() + (bary * gb).sum() + (dists * gd).sum()).backward()
        out.update({f"{tag}_verts": verts, f"{tag}_faces": faces, f"{tag}_persp": cfg["persp"], f"{tag}_blur": cfg["blur"],
                    f"{tag}_K": cfg["K"], f"{tag}_clipb": cfg["clipb"], f"{tag}_cull": cfg["cull"], f"{tag}_p2f": p2f,
                    f"{tag}_zbuf": zbuf, f"{tag}_bary": bary, f"{tag}_dists": dists, f"{tag}_gz": gz, f"{tag}_gb": gb,
                    f"{tag}_gd": gd, f"{tag}_grad_verts": verts.grad})
    mg.save("clip_ref", **out)


if __name__ == "__main__":
    main()
