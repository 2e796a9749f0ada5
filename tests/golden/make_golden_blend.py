"""Golden fixture for the blending row, generated FROM THE REFERENCE (build container only).

    python tests/golden/make_golden_blend.py   ->  tests/golden/blend_ref.npz

sigmoid_alpha_blend: the reference's C++ CPU kernels (oracle/_ref) forward + backward.
softmax_rgb_blend:   the reference's Python function (pytorch3d/renderer/blending.py:147-244) + torch autograd,
                     with scalar and per-batch-element znear / zfar.
"""
import os
import sys
from collections import namedtuple

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    import make_golden as mg

    ref = mg.bind_reference()
    from pytorch3d.renderer.blending import BlendParams, softmax_rgb_blend

    gen = torch.Generator().manual_seed(77)
    N, H, W, K = 2, 9, 7, 5
    p2f = torch.randint(-1, 40, (N, H, W, K), generator=gen)
    p2f[0, 0] = -1  # a fully empty row of pixels
    dists = (torch.rand(N, H, W, K, generator=gen) - 0.5) * 6e-4
    zbuf = torch.rand(N, H, W, K, generator=gen) * 4 + 0.5
    zbuf[p2f < 0] = -1
    dists[p2f < 0] = -1
    colors = torch.rand(N, H, W, K, 3, generator=gen)
    out = {"pix_to_face": p2f, "dists": dists, "zbuf": zbuf, "colors": colors}
    sigma = 1e-4
    alphas = ref.sigmoid_alpha_blend(dists.clone(), p2f.clone(), sigma)
    ga = torch.randn(N, H, W, generator=gen)
    gd = ref.sigmoid_alpha_blend_backward(ga, alphas, dists, p2f, sigma)
    out.update(sigma=sigma, sig_alphas=alphas, sig_grad_alphas=ga, sig_grad_dists=gd)

    Frag = namedtuple("Frag", "pix_to_face zbuf dists")
    cases = {"a": (1e-4, 1e-4, (1.0, 1.0, 1.0), 1.0, 100.0),
             "b": (3e-4, 5e-2, (0.1, 0.5, 0.9), torch.tensor([0.5, 1.0]), torch.tensor([20.0, 8.0]))}
    for tag, (sg, gm, bg, zn, zf) in cases.items():
        c = colors.clone().requires_grad_(True)
        d = dists.clone().requires_grad_(True)
        z = zbuf.clone().requires_grad_(True)
        img = softmax_rgb_blend(c, Frag(p2f, z, d), BlendParams(sigma=sg, gamma=gm, background_color=bg), znear=zn,
                                zfar=zf)
        g = torch.randn(img.shape, generator=gen)
        (img * g).sum().backward()
        out.update({f"sm_{tag}_sigma": sg, f"sm_{tag}_gamma": gm, f"sm_{tag}_bg": torch.tensor(bg),
                    f"sm_{tag}_znear": torch.as_tensor(zn, dtype=torch.float32),
                    f"sm_{tag}_zfar": torch.as_tensor(zf, dtype=torch.float32), f"sm_{tag}_out": img,
                    f"sm_{tag}_grad_out": g, f"sm_{tag}_grad_colors": c.grad, f"sm_{tag}_grad_dists": d.grad,
                    f"sm_{tag}_grad_zbuf": z.grad})
    mg.save("blend_ref", **out)


if __name__ == "__main__":
    main()
