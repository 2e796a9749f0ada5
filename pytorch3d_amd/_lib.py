"""ctypes binding of libp3d_amd.so (the C ABI declared in include/p3d_amd.h).

There is no fallback: if the HIP library is missing or cannot be loaded, every operator of this
package raises.  Build it with `python -m pytorch3d_amd.build` (hipcc, gfx950).
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# P3D_LIB_PATH selects an ablation build (profiles/ scripts only; see build.py)
LIB_PATH = os.environ.get("P3D_LIB_PATH") or os.path.join(_HERE, "libp3d_amd.so")

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_f32 = ctypes.c_float
c_ptr = ctypes.c_void_p
c_size = ctypes.c_size_t

_SIGNATURES = {
    # name: (restype, [argtypes])
    "p3d_abi_version": (c_int, []),
    "p3d_error_string": (ctypes.c_char_p, [c_int]),
    "p3d_rasterize_meshes_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_int, c_int, c_int]),
    "p3d_rasterize_meshes_short_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_int, c_int, c_int, c_i64]),
    "p3d_rasterize_meshes_workspace_need_offset": (c_size, [c_i64, c_int, c_int, c_int, c_int, c_int]),
    "p3d_rasterize_meshes": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_f32, c_int, c_int, c_int,
                                     c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "p3d_rasterize_meshes_naive": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_f32, c_int, c_int,
                                           c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "p3d_rasterize_meshes_coarse": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_f32, c_int, c_int, c_ptr,
                                            c_ptr, c_size, c_ptr]),
    "p3d_rasterize_fine_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int]),
    "p3d_rasterize_meshes_fine": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_f32,
                                          c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size,
                                          c_ptr]),
    "p3d_rasterize_meshes_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int,
                                              c_int, c_int, c_ptr, c_ptr]),
    "p3d_rasterize_meshes_backward_verts": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int,
                                                    c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "p3d_rasterize_meshes_cover_bytes": (c_size, [c_int, c_int, c_int]),
    "p3d_rasterize_meshes_cover_check": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "p3d_rasterize_meshes_with_cover": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_f32, c_int, c_int,
                                                c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size,
                                                c_ptr]),
    "p3d_rasterize_meshes_cover_list_bytes": (c_size, [c_int, c_int, c_int]),
    "p3d_rasterize_meshes_with_cover_list": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_f32, c_int, c_int,
                                                     c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size,
                                                     c_ptr]),
    "p3d_rasterize_meshes_backward_with_cover_list": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int,
                                                              c_int, c_int, c_int, c_ptr, c_ptr]),
    "p3d_rasterize_meshes_backward_verts_with_cover_list": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64,
                                                                    c_int, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "p3d_rasterize_meshes_cuda_order": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_f32, c_int, c_int,
                                                c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size,
                                                c_ptr]),
    "p3d_rasterize_meshes_backward_workspace_bytes": (c_size, [c_int, c_int, c_int]),
    "p3d_rasterize_meshes_backward_with_cover": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int,
                                                         c_int, c_int, c_int, c_ptr, c_ptr, c_size, c_ptr]),
    "p3d_rasterize_meshes_backward_verts_with_cover": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64,
                                                               c_int, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_size,
                                                               c_ptr]),
    "p3d_gather_face_verts": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    "p3d_gather_face_verts_pre": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    "p3d_rasterize_meshes_backward_pre": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_i64, c_int, c_int, c_int, c_int, c_int,
                                                  c_int, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "p3d_rasterize_meshes_backward_verts_pre": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64,
                                                        c_int, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "p3d_scatter_face_grads": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    "p3d_transform_gather_face_verts": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_ptr]),
    "p3d_transform_verts_forward": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr]),
    "p3d_transform_verts_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr]),
    "p3d_rasterize_points_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_int, c_int, c_int]),
    "p3d_rasterize_points_short_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_int, c_int, c_int, c_i64]),
    "p3d_rasterize_points_workspace_need_offset": (c_size, [c_i64, c_int, c_int, c_int, c_int, c_int]),
    "p3d_rasterize_points": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr,
                                     c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "p3d_rasterize_points_cuda_order": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr,
                                                c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "p3d_rasterize_points_naive": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_ptr, c_ptr,
                                           c_ptr, c_ptr]),
    "p3d_rasterize_points_coarse": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_int, c_ptr,
                                            c_ptr, c_size, c_ptr]),
    "p3d_rasterize_points_fine": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "p3d_rasterize_points_composite": (c_int, [c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                               ctypes.c_float, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "p3d_rasterize_points_composite_backward": (c_int, [c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_int,
                                                        ctypes.c_float, c_ptr, c_ptr, c_ptr]),
    "p3d_rasterize_points_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_ptr,
                                              c_ptr]),
    "p3d_composite_forward": (c_int, [c_int, c_ptr, c_ptr, c_ptr, c_int, c_int, c_i64, c_int, c_int, c_int,
                                      ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), c_ptr, c_ptr]),
    "p3d_composite_backward": (c_int, [c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_i64, c_int, c_int, c_int,
                                       ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), c_ptr, c_ptr, c_ptr]),
    "p3d_composite_forward_strided": (c_int, [c_int, c_ptr, ctypes.POINTER(c_i64), c_ptr, c_ptr, c_int, c_int, c_i64, c_int, c_int, c_int,
                                              ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), c_ptr, c_ptr]),
    "p3d_composite_backward_strided": (c_int, [c_int, c_ptr, c_ptr, ctypes.POINTER(c_i64), c_ptr, c_ptr, c_int, c_int, c_i64, c_int, c_int,
                                               c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), c_ptr, ctypes.POINTER(c_i64), c_ptr,
                                               c_ptr]),
    "p3d_interp_face_attrs_forward": (c_int, [c_int, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr]),
    "p3d_interp_face_attrs_backward": (c_int, [c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr,
                                               c_ptr]),
    "p3d_interp_face_attrs_backward_nhwk": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_i64, c_int,
                                                    c_ptr, c_ptr, c_ptr]),
    "p3d_clip_faces_plan_bytes": (c_size, [c_i64]),
    "p3d_clip_faces_plan": (c_int, [c_ptr, c_i64, ctypes.POINTER(c_f32), c_int, c_int, c_int, c_f32, c_ptr, c_size,
                                    c_ptr]),
    "p3d_clip_faces_emit": (c_int, [c_ptr, c_i64, c_ptr, c_int, c_ptr, c_size, c_i64, c_i64, c_i64, c_f32, c_int, c_ptr,
                                    c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "p3d_clip_faces_backward": (c_int, [c_ptr, c_i64, c_ptr, c_size, c_i64, c_i64, c_f32, c_int, c_ptr, c_ptr, c_ptr,
                                        c_ptr]),
    "p3d_convert_clipped_forward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr]),
    "p3d_convert_clipped_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr]),
    "p3d_sigmoid_alpha_blend_forward": (c_int, [c_ptr, c_ptr, c_f32, c_i64, c_int, c_ptr, c_ptr]),
    "p3d_sigmoid_alpha_blend_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_i64, c_int, c_ptr, c_ptr]),
    "p3d_softmax_rgb_blend_forward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_f32, ctypes.POINTER(c_f32), c_f32,
                                              c_f32, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr]),
    "p3d_softmax_rgb_blend_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_f32, ctypes.POINTER(c_f32),
                                               c_f32, c_f32, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr,
                                               c_ptr]),
    "p3d_phong_shade_forward": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_i64,
                                        c_ptr, c_ptr]),
    "p3d_phong_shade_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_int, c_int, c_int,
                                         c_int, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "p3d_soft_phong_supported_k": (c_int, [c_int]),
    "p3d_soft_phong_forward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_f32, c_f32,
                                       ctypes.POINTER(c_f32), c_f32, c_f32, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_i64,
                                       c_ptr, c_ptr]),
    "p3d_soft_phong_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_f32, c_f32,
                                        ctypes.POINTER(c_f32), c_f32, c_f32, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_i64,
                                        c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "p3d_sample_uv_forward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_i64, c_int, c_int, c_int,
                                      c_int, c_int, c_int, c_ptr, c_ptr]),
    "p3d_sample_uv_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_i64, c_int, c_int,
                                       c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr]),
    "p3d_sample_uv_multi_forward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_i64,
                                            c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    "p3d_sample_uv_multi_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int,
                                             c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr,
                                             c_ptr]),
    "p3d_sample_atlas_forward": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_ptr]),
    "p3d_sample_atlas_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_int, c_ptr, c_ptr]),
    "p3d_hard_rgb_blend_forward": (c_int, [c_ptr, c_ptr, ctypes.POINTER(c_f32), c_i64, c_int, c_ptr, c_ptr]),
    "p3d_hard_rgb_blend_backward": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr]),
    "p3d_profile_enable": (None, [c_int]),
    "p3d_profile_collect": (None, []),
    "p3d_profile_num_entries": (c_int, []),
    "p3d_profile_entry": (ctypes.c_char_p, [c_int, ctypes.POINTER(c_i64), ctypes.POINTER(ctypes.c_double)]),
    "p3d_profile_reset": (None, []),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lock = threading.Lock()
_lib = None


class ExtensionMissing(RuntimeError):
    pass


def load():
    """Load libp3d_amd.so; raise loudly when it is absent (there is no CPU or eager fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ExtensionMissing(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m pytorch3d_amd.build` "
                "(hipcc --offload-arch=gfx950). pytorch3d_amd has no CPU/eager fallback.")
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # e.g. libamdhip64 missing
            raise ExtensionMissing(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        if lib.p3d_abi_version() != 1:
            raise ExtensionMissing(f"{LIB_PATH}: ABI version {lib.p3d_abi_version()} != 1; rebuild")
        _lib = lib
    return _lib


def check(code, where):
    if code != 0:
        msg = load().p3d_error_string(code).decode()
        raise RuntimeError(f"{where}: {msg} (p3d error {code})")


def profile_snapshot():
    """{kernel name: (launches, total_ms)} since the last reset; synchronises the recorded events."""
    lib = load()
    lib.p3d_profile_collect()
    out = {}
    for i in range(lib.p3d_profile_num_entries()):
        n = c_i64(0)
        ms = ctypes.c_double(0.0)
        name = lib.p3d_profile_entry(i, ctypes.byref(n), ctypes.byref(ms))
        if name is not None and n.value > 0:
            out[name.decode()] = (n.value, ms.value)
    return out
