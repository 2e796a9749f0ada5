"""Build the pybind flavour of the drop-in boundary (pytorch3d_amd/csrc/bind.cpp) in-tree:

    python -m pytorch3d_amd.build_bind [--force]      ->  pytorch3d_amd/_C_pybind.so

A torch C++ extension (torch.utils.cpp_extension: g++ with the ROCm / ATen-HIP include paths, ninja) that links against
libp3d_amd.so (rpath $ORIGIN: the two files travel together).  Optional: the package's default boundary is the ctypes module
pytorch3d_amd/_C.py, which needs no compiler where the library is used; `shim.install(flavour="pybind")` selects this one.
"""
import glob
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_C_pybind.so")
SRC = os.path.join(HERE, "csrc", "bind.cpp")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")


def needs_build():
    if not os.path.exists(OUT):
        return True
    deps = [SRC, os.path.join(INCLUDE, "p3d_amd.h"), os.path.abspath(__file__)]
    return max(os.path.getmtime(p) for p in deps) > os.path.getmtime(OUT)


def build(force=False, verbose=False):
    if not (force or needs_build()):
        return OUT
    from . import build as lib_build

    lib_build.build()  # the library the module links against
    from torch.utils import cpp_extension

    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    bdir = os.path.join(HERE, "build", "_C_pybind")
    os.makedirs(bdir, exist_ok=True)
    cpp_extension.load(
        name="_C_pybind", sources=[SRC], extra_include_paths=[INCLUDE], build_directory=bdir, verbose=verbose, with_cuda=True,
        extra_cflags=["-O2", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"],
        # ($$: ninja's escape for the dollar of $ORIGIN; rpath: next to the library once copied in-tree; two levels up while it still lies in the build directory, where load() opens it)
        extra_ldflags=[f"-L{HERE}", "-l:libp3d_amd.so", "-Wl,-rpath,'$$ORIGIN'", "-Wl,-rpath,'$$ORIGIN/../..'"], is_python_module=True)
    built = glob.glob(os.path.join(bdir, "_C_pybind*.so"))
    if not built:
        raise RuntimeError("torch.utils.cpp_extension.load produced no _C_pybind*.so in " + bdir)
    shutil.copyfile(built[0], OUT)
    return OUT


def load():
    """The compiled module (built on first use where a compiler is at hand), or raises."""
    import importlib.util

    import torch  # noqa: F401  (libtorch must be loaded before the extension)

    path = OUT if (os.path.exists(OUT) and not needs_build()) else build()
    spec = importlib.util.spec_from_file_location("_C_pybind", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
