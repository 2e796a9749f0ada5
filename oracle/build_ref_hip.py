"""GPU-side reference checker (TEST INFRASTRUCTURE): the reference's own CUDA kernels of the hot path, built for gfx950.

    python oracle/build_ref_hip.py [--force]      ->  oracle/_ref/p3d_ref_hip.so   (build container only)

SURVEY.md 8c(3).  The reference's hot-path `.cu` files (rasterize_meshes, rasterize_coarse, rasterize_points, the three
compositors, interp_face_attrs, sigmoid_alpha_blend) + their CPU twins + oracle/ref_bind.cpp are copied to a scratch
directory OUTSIDE the repository (torch's hipify writes its translation next to the sources), translated by
torch.utils.hipify and compiled with hipcc for gfx950 with -DWITH_CUDA.  Only the resulting extension module lands in
oracle/_ref/ (git-ignored, shipped to the GPU box by gpurun).  It is used by tests (bit-level pin of the CUDA-order
oracle and of the HIP kernels against the reference's own device code) and by profiles/ref_hip_bench.py (same-GPU
timing of the reference's kernels).  The product never loads it; nothing here is a compatibility layer of the product.
"""
import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("P3D_REFERENCE_ROOT", "/root/reference")
REF_DIR = os.path.join(HERE, "_ref")
OUT_SO = os.path.join(REF_DIR, "p3d_ref_hip.so")              # hipcc defaults (-ffp-contract=fast, like nvcc's -fmad=true): timing, tolerance parity
OUT_SO_NOFMA = os.path.join(REF_DIR, "p3d_ref_hip_nofma.so")  # -ffp-contract=off: the expression order itself, bit for bit
SCRATCH = os.environ.get("P3D_REF_HIP_SCRATCH", "/tmp/p3d_ref_hip_build")

CU = [
    "rasterize_meshes/rasterize_meshes.cu", "rasterize_coarse/rasterize_coarse.cu", "rasterize_points/rasterize_points.cu",
    "compositing/alpha_composite.cu", "compositing/norm_weighted_sum.cu", "compositing/weighted_sum.cu",
    "interp_face_attrs/interp_face_attrs.cu", "blending/sigmoid_alpha_blend.cu",
    "face_areas_normals/face_areas_normals.cu", "packed_to_padded_tensor/packed_to_padded_tensor.cu",
]
CPP = [
    "rasterize_meshes/rasterize_meshes_cpu.cpp", "rasterize_points/rasterize_points_cpu.cpp",
    "blending/sigmoid_alpha_blend_cpu.cpp", "compositing/alpha_composite_cpu.cpp", "compositing/norm_weighted_sum_cpu.cpp",
    "compositing/weighted_sum_cpu.cpp", "face_areas_normals/face_areas_normals_cpu.cpp",
    "packed_to_padded_tensor/packed_to_padded_tensor_cpu.cpp",
]


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE, "pytorch3d", "csrc"))


def build(force=False, nofma=False):
    out_so = OUT_SO_NOFMA if nofma else OUT_SO
    name = "p3d_ref_hip_nofma" if nofma else "p3d_ref_hip"
    if not have_reference():
        return out_so if os.path.exists(out_so) else None
    if os.path.exists(out_so) and not force:
        return out_so
    csrc = os.path.join(REFERENCE, "pytorch3d", "csrc")
    if os.path.isdir(SCRATCH):
        shutil.rmtree(SCRATCH)
    shutil.copytree(csrc, os.path.join(SCRATCH, "csrc"), ignore=shutil.ignore_patterns("pulsar", "implicitron"))
    bind = os.path.join(SCRATCH, "csrc", "ref_bind_hip.cpp")
    src = open(os.path.join(HERE, "ref_bind.cpp")).read()
    # same names as the CPU module + the two operators that exist only on the device
    src = src.replace('  m.def("_rasterize_meshes_fine", &RasterizeMeshesFine);',
                      '  m.def("_rasterize_meshes_fine", &RasterizeMeshesFine);\n  m.def("_rasterize_points_fine", &RasterizePointsFine);\n'
                      '  m.def("interp_face_attrs_forward", &InterpFaceAttrsForward);\n  m.def("interp_face_attrs_backward", &InterpFaceAttrsBackward);')
    src = src.replace('#include "rasterize_points/rasterize_points.h"', '#include "rasterize_points/rasterize_points.h"\n#include "interp_face_attrs/interp_face_attrs.h"')
    if nofma:
        src = src.replace("TORCH_EXTENSION_NAME", "p3d_ref_hip_nofma")
    open(bind, "w").write(src)
    os.makedirs(REF_DIR, exist_ok=True)
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    import torch
    from torch.utils import cpp_extension as ce

    sources = [os.path.join(SCRATCH, "csrc", s) for s in CU + CPP] + [bind]
    build_dir = os.path.join(SCRATCH, "build_" + name)
    os.makedirs(build_dir, exist_ok=True)
    dev_flags = ["-DWITH_CUDA", "-O3", "-w", "--offload-arch=gfx950"] + (["-ffp-contract=off"] if nofma else [])
    ce.load(name=name, sources=sources, extra_include_paths=[os.path.join(SCRATCH, "csrc")],
            extra_cflags=["-DWITH_CUDA", "-O2", "-w"], extra_cuda_cflags=dev_flags,
            build_directory=build_dir, verbose=False, is_python_module=False, with_cuda=True)
    shutil.copy2(os.path.join(build_dir, name + ".so"), out_so)
    return out_so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build(force="--force" in sys.argv, nofma=True))
