import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_soft_phong as T
import pytorch3d_amd as p3d, pytorch3d_amd.shading as sh
from oracle import oracle as orc
for K in (1, 2):
    d, m, frag, world, normals, gen, B = T._scene(K, seed=K)
    N, H, W, _ = frag.pix_to_face.shape
    L = sh.Lights(torch.full((1, 3), 0.4, device=d), torch.full((1, 3), 0.6, device=d), torch.full((1, 3), 0.3, device=d), location=torch.tensor([[0.5, 1.0, -2.0]], device=d))
    M = sh.Materials(torch.ones(1, 3, device=d), torch.ones(1, 3, device=d), torch.ones(1, 3, device=d), torch.tensor([12.0], device=d))
    cam = T.Cam(torch.tensor([[0., 0., -3.]], device=d))
    vcol = torch.rand(world.shape[0], 3, generator=gen).to(d)
    mesh = T._Mesh(world, normals, m.faces_packed())
    bp = p3d.BlendParams(sigma=1e-3, gamma=1e-2, background_color=(0.2, 0.5, 0.9))
    g_img = torch.randn((N, H, W, 4), generator=gen).to(d)
    z1 = frag.zbuf.clone().requires_grad_(True); z2 = frag.zbuf.clone().requires_grad_(True)
    fr1 = T.Frag(frag.pix_to_face, z1, frag.bary_coords, frag.dists); fr2 = T.Frag(frag.pix_to_face, z2, frag.bary_coords, frag.dists)
    colors = sh.phong_shading_vertex_colors(mesh, fr1, L, cam, M, vcol)
    p3d.softmax_rgb_blend(colors, fr1, bp, znear=0.5, zfar=6.0).backward(g_img)
    sh.soft_phong_shading(mesh, fr2, L, cam, M, None, bp, znear=0.5, zfar=6.0, verts_colors_packed=vcol).backward(g_img)
    # oracle for the blend backward
    o = orc.softmax_rgb_blend_backward(g_img.cpu(), colors.detach().cpu(), frag.pix_to_face.cpu(), frag.dists.cpu(), frag.zbuf.cpu(), 1e-3, 1e-2, (0.2, 0.5, 0.9), 0.5, 6.0) if hasattr(orc, "softmax_rgb_blend_backward") else None
    valid = frag.pix_to_face >= 0
    idx = valid.nonzero()[:5]
    print("K", K, "unfused gz", z1.grad[valid][:5].tolist(), "fused gz", z2.grad[valid][:5].tolist())
    if o is not None:
        print("   oracle gz", o[2][valid.cpu()][:5].tolist())
    print("   max abs unfused", float(z1.grad.abs().max()), "fused", float(z2.grad.abs().max()))
