"""Install pytorch3d_amd._C as `pytorch3d._C`, so the UNMODIFIED reference Python package
(pytorch3d.renderer.MeshRasterizer, PointsRasterizer, AlphaCompositor, NormWeightedCompositor,
interpolate_face_attributes, ...) runs on the MI355X kernels.

    import pytorch3d_amd.shim as shim
    shim.install()                       # reference already importable, or
    shim.install("/path/to/pytorch3d")   # a source checkout (no build needed: only _C is native)
    from pytorch3d.renderer import MeshRasterizer   # drop-in

`pytorch3d/__init__.py` does not import `_C`; the sub-packages do (`from pytorch3d import _C`),
so the module must be in sys.modules before they are imported.  Operators outside the hot path
(knn, point_mesh, pulsar, ...) raise NotImplementedError when called.
"""
import sys
import types

from . import _C as _ours


def make_module():
    mod = types.ModuleType("pytorch3d._C")
    mod.__doc__ = "pytorch3d._C provided by pytorch3d_amd (MI355X rasterization hot path)"
    for name in _ours.HOT_PATH_EXPORTS:
        setattr(mod, name, getattr(_ours, name))
    for name in ("EPS", "MAX_FLOAT", "MAX_INT", "MAX_UINT", "MAX_USHORT", "PULSAR_MAX_GRAD_SPHERES"):
        setattr(mod, name, getattr(_ours, name))

    def __getattr__(name):  # PEP 562: anything else is outside the hot path
        if name.startswith("__"):
            raise AttributeError(name)

        def _missing(*args, **kwargs):
            raise NotImplementedError(f"pytorch3d._C.{name} is outside the rasterization hot path that "
                                      "pytorch3d_amd implements")

        return _missing

    mod.__getattr__ = __getattr__
    return mod


def install(reference_root=None):
    """Register the shim (idempotent).  Returns the module object."""
    if reference_root is not None and reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    existing = sys.modules.get("pytorch3d._C")
    if existing is not None and getattr(existing, "__p3d_amd__", False):
        return existing
    mod = make_module()
    mod.__p3d_amd__ = True
    sys.modules["pytorch3d._C"] = mod
    pkg = sys.modules.get("pytorch3d")
    if pkg is not None:
        setattr(pkg, "_C", mod)
    return mod
