"""BASELINE configs[1] as specified (SURVEY.md §8d "config 2"): the reference's cow, generated FROM THE REFERENCE
(build container only):

    python tests/golden/make_golden_cow.py   ->  tests/golden/cow_ref.npz

`docs/tutorials/data/cow_mesh/cow.obj` (2930 verts / 5856 faces; read with a 10-line OBJ reader because pytorch3d.io
needs iopath), `FoVPerspectiveCameras` at `look_at_view_transform(2.7, 0, 180)`, the reference's own
`MeshRasterizer` (renderer/mesh/rasterizer.py:139-260) on CPU = its C++ CPU rasterizer (oracle/_ref): 256x256,
faces_per_pixel=8, blur_radius=1e-4, perspective_correct, clip_barycentric_coords, no culling, no z-clipping;
plus its CPU backward for seeded upstream gradients.  Stored with it: the world vertices, the vertices in NDC as
`MeshRasterizer.transform` produces them and the two 4x4 matrices (row-vector convention) that take world -> view ->
NDC, for the fused world->NDC entry point.  tests/test_gpu_baseline_sizes.py replays it on the HIP kernels.
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


def read_obj(path):
    v, f = [], []
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                v.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                idx = [int(x.split("/")[0]) - 1 for x in t[1:]]
                for k in range(1, len(idx) - 1):
                    f.append([idx[0], idx[k], idx[k + 1]])
    return torch.tensor(v, dtype=torch.float32), torch.tensor(f, dtype=torch.int64)


def main():
    import make_golden as mg

    ref = mg.bind_reference()
    from pytorch3d.renderer import FoVPerspectiveCameras, MeshRasterizer, RasterizationSettings, look_at_view_transform
    from pytorch3d.structures import Meshes

    verts, faces = read_obj(os.path.join(mg.REFERENCE, "docs", "tutorials", "data", "cow_mesh", "cow.obj"))
    assert verts.shape == (2930, 3) and faces.shape == (5856, 3), (verts.shape, faces.shape)
    meshes = Meshes(verts=[verts], faces=[faces])
    R, T = look_at_view_transform(2.7, 0, 180)
    cameras = FoVPerspectiveCameras(R=R, T=T)
    H = W = 256
    K = 8
    blur = 1e-4
    settings = RasterizationSettings(image_size=H, blur_radius=blur, faces_per_pixel=K, perspective_correct=True,
                                     clip_barycentric_coords=True, cull_backfaces=False, bin_size=0,
                                     z_clip_value=None, cull_to_frustum=False)
    rasterizer = MeshRasterizer(cameras=cameras, raster_settings=settings)
    torch.set_num_threads(os.cpu_count() or 1)
    fr = rasterizer(meshes)
    ndc = rasterizer.transform(meshes).verts_packed()
    w2v = cameras.get_world_to_view_transform().get_matrix()[0]
    proj = cameras.get_projection_transform().get_matrix()[0]
    fv = ndc[faces].contiguous()
    gen = torch.Generator().manual_seed(231)
    gz = torch.randn(fr.zbuf.shape, generator=gen)
    gb = torch.randn(fr.bary_coords.shape, generator=gen)
    gd = torch.randn(fr.dists.shape, generator=gen)
    gfv = ref.rasterize_meshes_backward(fv, fr.pix_to_face, gz, gb, gd, True, True)
    cover = float((fr.pix_to_face[..., 0] >= 0).float().mean())
    hits = float((fr.pix_to_face >= 0).float().sum() / (H * W))
    print(f"coverage {cover:.3f}, hits/pixel {hits:.2f}")
    mg.save("cow_ref", verts_world=verts, faces=faces.to(torch.int32), verts_ndc=ndc, world_to_view=w2v, projection=proj,
            pix_to_face=fr.pix_to_face.to(torch.int32), zbuf=fr.zbuf, bary=fr.bary_coords, dists=fr.dists,
            grad_face_verts=gfv, image_size=H, K=K, blur_radius=blur, seed=231)


if __name__ == "__main__":
    main()
