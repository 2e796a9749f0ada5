import math, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import run_reference_suite as rrs
rrs._stub_missing_packages()
import pytorch3d_amd.shim as shim
shim.install(os.path.join(ROOT, "oracle", "_ref", "reference_py"))
from pytorch3d.renderer import FoVOrthographicCameras, MeshRasterizer, RasterizationSettings
from pytorch3d.structures import Meshes
if len(sys.argv) > 1 and sys.argv[1] == "patched":
    shim.patch_reference_python()
import _util as U
d = torch.device("cuda:0")
B, H, K = 64, 512, 8
blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
verts, faces = U.hetero_batch(B, seed=0)
mesh0 = Meshes(verts=[v.to(d) for v in verts], faces=[f.to(d) for f in faces])
V = int(mesh0.verts_packed().shape[0])
deform = torch.zeros((V, 3), device=d, requires_grad=True)
cams = FoVOrthographicCameras(device=d)
rs = RasterizationSettings(image_size=H, blur_radius=blur, faces_per_pixel=K, perspective_correct=True, clip_barycentric_coords=True)
rast = MeshRasterizer(cameras=cams, raster_settings=rs)
gen = torch.Generator().manual_seed(231)
g = [torch.randn(s, generator=gen).to(d) for s in ((B, H, H, K), (B, H, H, K, 3), (B, H, H, K))]
from pytorch3d.renderer.mesh import rasterize_meshes as rm_mod
import pytorch3d.renderer.mesh.rasterizer as rz
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
for it in range(8):
    if it == 3: T = {}
    deform.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = mesh0.offset_verts(deform); tick("offset_verts", t0)
    t0 = time.perf_counter(); mp = rast.transform(m); tick("transform", t0)
    t0 = time.perf_counter()
    out = rz.rasterize_meshes(mp, image_size=H, blur_radius=blur, faces_per_pixel=K, bin_size=None, max_faces_per_bin=None,
                              clip_barycentric_coords=True, perspective_correct=True, cull_backfaces=False, z_clip_value=0.5, cull_to_frustum=False)
    tick("rasterize_meshes (L2 function)", t0)
    t0 = time.perf_counter(); torch.autograd.backward(list(out[1:]), g); tick("backward (all)", t0)
print(json.dumps({k: round(v / 5 * 1e3, 3) for k, v in T.items()}))
