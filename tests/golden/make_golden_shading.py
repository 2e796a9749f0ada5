"""Golden fixture for the shading row, generated FROM THE REFERENCE (build container only).

    python tests/golden/make_golden_shading.py   ->  tests/golden/shading_ref.npz

The reference's own `phong_shading` (pytorch3d/renderer/mesh/shading.py:99-112) with its own PointLights /
DirectionalLights / AmbientLights / Materials classes (renderer/lighting.py, renderer/materials.py) and torch
autograd, on fragments rasterized by the reference's C++ CPU kernel.  The mesh and camera arguments are thin
objects exposing exactly the methods phong_shading calls (verts_packed / faces_packed / verts_normals_packed,
get_camera_center), so that the gradients wrt vertex positions and vertex normals are recorded separately.
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


class MeshStub:
    def __init__(self, verts, faces, normals):
        self.v, self.f, self.n = verts, faces, normals

    def verts_packed(self):
        return self.v

    def faces_packed(self):
        return self.f

    def verts_normals_packed(self):
        return self.n

    def faces_normals_packed(self):  # Meshes.faces_normals_packed needs _C.face_areas_normals (not a hot-path operator):
        fv = self.v[self.f]          # face_areas_normals_cpu.cpp:44-62 restated with torch ops
        c = torch.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0], dim=1)
        return c / c.norm(dim=1, keepdim=True).clamp_min(1e-6)


class CamStub:
    def __init__(self, c):
        self.c = c

    def clone(self):
        return CamStub(self.c.clone())

    def gather_props(self, idx):  # TensorProperties.gather_props for the one property phong / gouraud shading read
        self.c = self.c[idx]
        return self

    def get_camera_center(self):
        return self.c


def main():
    import make_golden as mg
    import _util as U

    ref = mg.bind_reference()
    from pytorch3d.ops import interpolate_face_attributes
    from pytorch3d.renderer.lighting import AmbientLights, DirectionalLights, PointLights
    from pytorch3d.renderer.materials import Materials
    from pytorch3d.renderer.mesh.shading import flat_shading, gouraud_shading, phong_shading
    from pytorch3d.renderer.mesh.textures import TexturesVertex
    from pytorch3d.structures import Meshes

    gen = torch.Generator().manual_seed(404)
    verts_l, faces_l = U.hetero_batch(2, seed=5, fmin=150, fmax=400)
    real = Meshes(verts=verts_l, faces=faces_l)
    verts, faces = real.verts_packed().clone(), real.faces_packed().clone()
    normals = real.verts_normals_packed().clone()
    fv = verts[faces]
    N, H, W, K = 2, 24, 20, 3
    p2f, zbuf, bary, dists = ref.rasterize_meshes(fv, real.mesh_to_faces_packed_first_idx(), real.num_faces_per_mesh(),
                                                  torch.full((fv.shape[0],), -1, dtype=torch.int64), (H, W), 1e-4, K, 0,
                                                  0, True, True, False)
    Frag = namedtuple("Frag", "pix_to_face bary_coords")
    out = {"verts": verts, "faces": faces, "normals": normals, "pix_to_face": p2f, "bary": bary}
    cam = torch.tensor([[0.3, -0.2, -2.5], [-1.0, 0.5, -3.0]])
    colors_v = torch.rand(verts.shape[0], 3, generator=gen)
    texels0 = torch.rand(N, H, W, K, 3, generator=gen)
    out.update(camera_center=cam, verts_colors=colors_v, texels=texels0)
    lights = {
        "point": PointLights(ambient_color=((0.4, 0.5, 0.3),), diffuse_color=((0.6, 0.2, 0.7), (0.3, 0.9, 0.5)),
                             specular_color=((0.5, 0.7, 0.2),), location=((0.5, 1.5, -2.0), (-1.0, 2.0, -1.0))),
        "dir": DirectionalLights(ambient_color=((0.2, 0.2, 0.6), (0.5, 0.1, 0.1)), diffuse_color=((0.8, 0.7, 0.6),),
                                 specular_color=((0.3, 0.3, 0.9),), direction=((0.2, 1.0, -0.7),)),
        "amb": AmbientLights(ambient_color=((0.9, 0.6, 0.3),)),
    }
    mats = {
        "point": Materials(ambient_color=((0.9, 0.8, 0.7),), diffuse_color=((0.5, 0.6, 0.7),),
                           specular_color=((0.8, 0.9, 1.0),), shininess=6.0),  # batch 1: the reference cannot broadcast (N,3) materials
        "dir": Materials(shininess=64),
        "amb": Materials(ambient_color=((0.5, 0.5, 1.0),)),
    }
    for tag in ("point", "dir", "amb"):
        L, M = lights[tag], mats[tag]
        for kind in ("texels", "vcol"):
            v = verts.clone().requires_grad_(True)
            nrm = normals.clone().requires_grad_(True)
            b = bary.clone().requires_grad_(True)
            if kind == "texels":
                t_in = texels0.clone().requires_grad_(True)
                texels = t_in
            else:
                t_in = colors_v.clone().requires_grad_(True)
                texels = interpolate_face_attributes(p2f, b, t_in[faces])
            col = phong_shading(MeshStub(v, faces, nrm), Frag(p2f, b), L, CamStub(cam), M, texels)
            g = torch.randn(col.shape, generator=gen)
            (col * g).sum().backward()
            pre = f"{tag}_{kind}_"
            rec = {pre + "colors": col, pre + "grad_colors": g, pre + "grad_verts": v.grad, pre + "grad_normals": nrm.grad,
                   pre + "grad_bary": b.grad, pre + "grad_tex": t_in.grad}
            out.update({k: x for k, x in rec.items() if x is not None})  # ambient-only lights: no gradient to the geometry
        for n_ in ("ambient_color", "diffuse_color", "specular_color", "location", "direction"):
            if hasattr(L, n_) and torch.is_tensor(getattr(L, n_)):
                out[f"{tag}_light_{n_}"] = getattr(L, n_)
        for n_ in ("ambient_color", "diffuse_color", "specular_color", "shininess"):
            out[f"{tag}_mat_{n_}"] = getattr(M, n_)
    # flat shading (shading.py:178-225) and Gouraud shading (shading.py:125-175, through the real Meshes + TexturesVertex)
    nv = [int(x) for x in real.num_verts_per_mesh()]
    for tag in ("point", "dir"):
        L, M = lights[tag], mats[tag]
        v = verts.clone().requires_grad_(True)
        t_in = texels0.clone().requires_grad_(True)
        col = flat_shading(MeshStub(v, faces, None), Frag(p2f, bary), L, CamStub(cam), M, t_in)
        g = torch.randn(col.shape, generator=gen)
        (col * g).sum().backward()
        out.update({f"{tag}_flat_colors": col, f"{tag}_flat_grad_colors": g, f"{tag}_flat_grad_verts": v.grad,
                    f"{tag}_flat_grad_tex": t_in.grad})
        vl = [x.clone().requires_grad_(True) for x in verts.split(nv)]
        cl = [x.clone().requires_grad_(True) for x in colors_v.split(nv)]
        b = bary.clone().requires_grad_(True)
        gm = Meshes(verts=vl, faces=faces_l, textures=TexturesVertex(verts_features=cl))
        col = gouraud_shading(gm, Frag(p2f, b), L, CamStub(cam), M)
        g = torch.randn(col.shape, generator=gen)
        (col * g).sum().backward()
        out.update({f"{tag}_gouraud_colors": col, f"{tag}_gouraud_grad_colors": g,
                    f"{tag}_gouraud_grad_verts": torch.cat([x.grad for x in vl]),
                    f"{tag}_gouraud_grad_tex": torch.cat([x.grad for x in cl]), f"{tag}_gouraud_grad_bary": b.grad})
    out["num_verts_per_mesh"] = torch.tensor(nv)
    # gradients wrt the lights, the materials and the camera centre (phong_shading, texels given)
    for tag, cls, vec_name in (("point", PointLights, "location"), ("dir", DirectionalLights, "direction")):
        L0, M0 = lights[tag], mats[tag]
        leaf = lambda t: t.detach().clone().requires_grad_(True)
        lt = {n_: leaf(getattr(L0, n_)) for n_ in ("ambient_color", "diffuse_color", "specular_color", vec_name)}
        mt = {n_: leaf(getattr(M0, n_)) for n_ in ("ambient_color", "diffuse_color", "specular_color", "shininess")}
        camt = leaf(cam)
        L = cls(**lt)
        M = Materials(**mt)
        col = phong_shading(MeshStub(verts, faces, normals), Frag(p2f, bary), L, CamStub(camt), M, texels0)
        g = torch.randn(col.shape, generator=gen)
        (col * g).sum().backward()
        pre = f"{tag}_pg_"
        out.update({pre + "colors": col, pre + "grad_colors": g, pre + "grad_camera": camt.grad})
        out.update({pre + "grad_light_" + k: v.grad for k, v in lt.items()})
        out.update({pre + "grad_mat_" + k: v.grad for k, v in mt.items()})
    mg.save("shading_ref", **out)
    print({k: tuple(v.shape) for k, v in out.items() if k.startswith("point_texels")})
    print("coverage", float((p2f >= 0).float().mean()))


if __name__ == "__main__":
    main()
