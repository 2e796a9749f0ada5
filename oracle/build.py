"""Build recipes for the test oracle (TEST INFRASTRUCTURE, not product code).

build_oracle()  gcc  oracle/p3d_oracle.c            -> oracle/libp3d_oracle.so
build_ref()     g++  reference hot-path CPU sources -> oracle/_ref/p3d_ref_cpu.so
                (only where /root/reference exists; compiled from the sources
                where they lie, nothing is copied into this repository)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("P3D_REFERENCE_ROOT", "/root/reference")
ORACLE_SO = os.path.join(HERE, "libp3d_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")
REF_SO = os.path.join(REF_DIR, "p3d_ref_cpu.so")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def build_oracle(force=False):
    src = os.path.join(HERE, "p3d_oracle.c")
    if force or _stale(ORACLE_SO, [src]):
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
               "-fvisibility=hidden", src, "-o", ORACLE_SO, "-lm"]
        subprocess.check_call(cmd)
    return ORACLE_SO


_REF_SOURCES = [
    "pytorch3d/csrc/rasterize_meshes/rasterize_meshes_cpu.cpp",
    "pytorch3d/csrc/rasterize_points/rasterize_points_cpu.cpp",
    "pytorch3d/csrc/blending/sigmoid_alpha_blend_cpu.cpp",
    "pytorch3d/csrc/compositing/alpha_composite_cpu.cpp",
    "pytorch3d/csrc/compositing/norm_weighted_sum_cpu.cpp",
    "pytorch3d/csrc/compositing/weighted_sum_cpu.cpp",
    # not on the hot path: the reference's own unit tests (tests/run_reference_suite.py) reach them through
    # Meshes.faces_normals_packed / packed_to_padded, so the test runner routes them to the reference's CPU code
    "pytorch3d/csrc/face_areas_normals/face_areas_normals_cpu.cpp",
    "pytorch3d/csrc/packed_to_padded_tensor/packed_to_padded_tensor_cpu.cpp",
]


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE, "pytorch3d", "csrc"))


def build_ref(force=False):
    """Compile the reference's own CPU implementation of the hot path (no CUDA) into
    oracle/_ref/p3d_ref_cpu.so, a torch extension exporting the reference's pybind names."""
    if not have_reference():
        return REF_SO if os.path.exists(REF_SO) else None
    bind = os.path.join(HERE, "ref_bind.cpp")
    srcs = [os.path.join(REFERENCE, s) for s in _REF_SOURCES] + [bind]
    if not (force or _stale(REF_SO, srcs)):
        return REF_SO
    os.makedirs(REF_DIR, exist_ok=True)
    import torch
    from torch.utils import cpp_extension as ce
    import sysconfig

    inc = ["-I" + os.path.join(REFERENCE, "pytorch3d", "csrc")]
    inc += ["-I" + p for p in ce.include_paths()]
    inc += ["-I" + sysconfig.get_paths()["include"]]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cxx11 = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    objs = []
    for s in srcs:
        o = os.path.join(REF_DIR, os.path.basename(s) + ".o")
        cmd = ["g++", "-O2", "-fPIC", "-std=c++17", "-c", s, "-o", o, "-DTORCH_EXTENSION_NAME=p3d_ref_cpu",
               "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={cxx11}", "-w"] + inc
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = ["g++", "-shared", "-o", REF_SO] + objs + ["-L" + libdir, "-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python",
                                                     "-Wl,-rpath," + libdir]
    subprocess.check_call(cmd)
    for o in objs:
        os.remove(o)
    return REF_SO


if __name__ == "__main__":
    print(build_oracle(force="--force" in sys.argv))
    print(build_ref(force="--force" in sys.argv))
