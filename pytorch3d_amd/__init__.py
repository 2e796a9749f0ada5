"""pytorch3d_amd -- PyTorch3D's differentiable-rasterization hot path, re-implemented for
AMD MI355X (CDNA4 / gfx950) as hand-written HIP behind a C ABI (include/p3d_amd.h).

    pytorch3d_amd._C                 the `pytorch3d._C` operator surface (drop-in boundary)
    pytorch3d_amd.shim.install()     register it as pytorch3d._C for the unmodified reference
    rasterize_meshes, rasterize_points, alpha_composite, norm_weighted_sum, weighted_sum,
    interpolate_face_attributes      host-side mirrors of the reference's L2 functions
    clip_faces, softmax_rgb_blend, sigmoid_alpha_blend, hard_rgb_blend, phong_shading, sample_textures_uv,
    sample_textures_atlas            the neighbouring steps, fused

Importing the package does not load the HIP library; the first operator call does, and raises
if it is missing (no CPU / eager fallback exists).
"""
from . import _C  # noqa: F401
from .blending import BlendParams, hard_rgb_blend, sigmoid_alpha_blend, softmax_rgb_blend  # noqa: F401
from .compositing import alpha_composite, norm_weighted_sum, weighted_sum  # noqa: F401
from .interp_face_attrs import interpolate_face_attributes  # noqa: F401
from .rasterize_meshes import rasterize_meshes, rasterize_meshes_world  # noqa: F401
from .rasterize_points import rasterize_points  # noqa: F401
from .render_points import render_points_alpha  # noqa: F401
from .shading import (flat_shading, gouraud_shading, phong_shading, phong_shading_vertex_colors,  # noqa: F401
                      soft_phong_shading)
from .structures import PackedMeshes, PackedPointclouds  # noqa: F401
from .textures import sample_textures_atlas, sample_textures_uv  # noqa: F401

__version__ = "0.2.0"
