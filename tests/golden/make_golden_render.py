"""End-to-end golden for the chain rasterize -> Phong shading -> softmax blend, generated FROM THE REFERENCE (build
container only):

    python tests/golden/make_golden_render.py   ->  tests/golden/render_ref.npz, render_ref_k8.npz

The reference's own `MeshRenderer(MeshRasterizer, SoftPhongShader)` (renderer/mesh/renderer.py:41-63, rasterizer.py:139-260,
shader.py SoftPhongShader) with FoVPerspectiveCameras, PointLights, Materials and TexturesVertex on CPU (its C++ CPU
rasterizer + its Python shading / blending), plus torch autograd of a random loss to the vertices and vertex colours.
Stored next to the image: everything the renderer derives on the way that is OUT of this repository's scope (camera
transforms), i.e. the vertices in NDC as MeshRasterizer.transform produces them, the camera centre, znear / zfar.
tests/test_gpu_render_chain.py replays the chain through pytorch3d_amd's fused kernels.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(K=6, name="render_ref"):
    import make_golden as mg
    import _util as U

    mg.bind_reference()
    from pytorch3d.renderer import (BlendParams, FoVPerspectiveCameras, Materials, MeshRasterizer, MeshRenderer,
                                    PointLights, RasterizationSettings, SoftPhongShader, TexturesVertex,
                                    look_at_view_transform)
    from pytorch3d.structures import Meshes

    gen = torch.Generator().manual_seed(11)
    v0, f0 = U.ico_sphere(2)
    v1, f1 = U.torus(0.35, 0.9, 10, 14)
    verts_l = [v0.clone().requires_grad_(True), (v1 * 0.9).clone().requires_grad_(True)]
    faces_l = [f0, f1]
    cols_l = [torch.rand(v.shape[0], 3, generator=gen).requires_grad_(True) for v in verts_l]
    meshes = Meshes(verts=verts_l, faces=faces_l, textures=TexturesVertex(verts_features=cols_l))
    R, T = look_at_view_transform(dist=2.7, elev=torch.tensor([10.0, 35.0]), azim=torch.tensor([20.0, -50.0]))
    cameras = FoVPerspectiveCameras(R=R, T=T, znear=1.0, zfar=100.0)
    sigma, gamma = 1e-4, 1e-4
    import math

    settings = RasterizationSettings(image_size=48, blur_radius=math.log(1.0 / 1e-4 - 1.0) * sigma, faces_per_pixel=K,
                                     perspective_correct=True, clip_barycentric_coords=True, cull_backfaces=False,
                                     bin_size=0)
    lights = PointLights(location=((1.5, 2.0, -2.0), (-2.0, 1.0, -1.5)), ambient_color=((0.4, 0.4, 0.4),),
                         diffuse_color=((0.5, 0.4, 0.6),), specular_color=((0.3, 0.3, 0.3),))
    materials = Materials(shininess=24.0)
    blend = BlendParams(sigma=sigma, gamma=gamma, background_color=(0.2, 0.3, 0.4))
    rasterizer = MeshRasterizer(cameras=cameras, raster_settings=settings)
    renderer = MeshRenderer(rasterizer, SoftPhongShader(cameras=cameras, lights=lights, materials=materials, blend_params=blend))
    img = renderer(meshes)
    g = torch.randn(img.shape, generator=gen)
    (img * g).sum().backward()
    fr = rasterizer(meshes)
    ndc = rasterizer.transform(meshes).verts_packed()
    out = {"image": img, "grad_image": g, "verts_world": meshes.verts_packed(), "faces": meshes.faces_packed(),
           "verts_ndc": ndc, "verts_colors": torch.cat(cols_l), "num_verts": torch.tensor([v.shape[0] for v in verts_l]),
           "num_faces": torch.tensor([f.shape[0] for f in faces_l]), "camera_center": cameras.get_camera_center(),
           "light_location": lights.location, "light_ambient": lights.ambient_color, "light_diffuse": lights.diffuse_color,
           "light_specular": lights.specular_color, "shininess": materials.shininess, "sigma": sigma, "gamma": gamma,
           "background": torch.tensor(blend.background_color), "blur_radius": settings.blur_radius, "K": K, "image_size": 48,
           "znear": 1.0, "zfar": 100.0, "pix_to_face": fr.pix_to_face, "zbuf": fr.zbuf,
           "grad_verts_colors": torch.cat([c.grad for c in cols_l]),
           # gradient wrt the NDC vertices is not separable from the camera transform here; the colour gradient and
           # the image pin the chain, vertex gradients are covered per stage by the other fixtures
           }
    mg.save(name, **out)
    print("coverage", float((fr.pix_to_face[..., 0] >= 0).float().mean()))


if __name__ == "__main__":
    main()
    # the same scene at K = 8: one of the capacities of the fused soft-Phong kernels (csrc/soft_phong.hip: 1, 2, 4, 8, 16)
    main(K=8, name="render_ref_k8")
