"""Build libp3d_amd.so (the C-ABI HIP library, include/p3d_amd.h) in-tree with hipcc for gfx950.

    python -m pytorch3d_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the arithmetic contract
(bit-exact pix_to_face needs the reference's expression trees without FMA contraction);
-munsafe-fp-atomics selects the hardware global_atomic_add_f32 for the backward scatters;
-fno-slp-vectorize keeps scalar f32 code out of v_pk_* (the operand shuffles cost more than the packing saves).
"""
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# P3D_LIB_PATH: build / load an ABLATION variant next to the product library (profiles/ only; the driver never sets it)
LIB = os.environ.get("P3D_LIB_PATH") or os.path.join(HERE, "libp3d_amd.so")
ARCH = "gfx950"

SOURCES = ["binning.hip", "raster_mesh.hip", "raster_mesh_bwd.hip", "gather.hip", "transform.hip", "raster_points.hip", "render_points.hip", "composite.hip", "blend.hip", "clip.hip", "interp.hip", "shade.hip", "soft_phong.hip", "texture.hip", "texture_multi.hip", "atlas.hip", "profile.cpp"]
HEADERS = ["binning.h", "p3d_common.h", "p3d_geom.h", "topk.h", "topk_insert_asm.h", "wave_table.h", "tile_map.h", "chunk_order.h", "atlas_cell.h", "uvm_sample.h", "shade_sample.h", os.path.join("..", "..", "include", "p3d_amd.h")]

FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off",
    "-munsafe-fp-atomics", "-Wno-unused-result",
    # hipcc's SLP vectorizer packs adjacent scalar f32 ops into v_pk_* and pays for it with register shuffles
    # (1000+ v_mov in the unrolled backward): measured -7% on mesh_backward, -3.5% on mesh_fine without it
    "-fno-slp-vectorize",
]


# Kernels that keep registers in AGPRs (more than 256 live VGPRs at one wave per SIMD).  Round 4 saw kernels of raster_mesh.hip
# LOSE queue entries whenever VGPRs left the register file inside their candidate loop -- scratch spills, or AGPR copies
# (profiles/r04/spill_miscompile.md).  Round 6 found why (profiles/r06/spill_root_cause.md): the kernels read lane-indexed tables with
# v_readlane inside partially active regions; a spill reload there restores the ACTIVE lanes only.  VGPR spills are refused outright; an AGPR
# kernel is accepted only if it is listed here with the GPU test that runs it on inputs that make EVERY register row live.
AGPR_KERNELS_TESTED = {
    "softmax_blend_bwd_kernel<32>": "tests/test_gpu_blending.py::test_blend_kernels_vs_oracle_all_capacities[17|24|32-dense]",
    "composite_bwd_tile_kernel<0, 32>": "tests/test_gpu_points_composite_interp.py::test_compositors[17|24|32]",
}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths if os.path.exists(p))


def _extra_flags():
    return os.environ.get("P3D_EXTRA_FLAGS", "").split()  # flags of a temporary probe / experiment build (always into a separate library: P3D_LIB_PATH)


def _flag_signature():
    return " ".join(FLAGS + _extra_flags())


def _stamp_path():
    return LIB + ".flags"


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(LIB + ".resources.json"):  # the library and its resource record go together
        return True
    # a library built with other flags (ablation -D switches, table sizes) must never be reused as the product build
    try:
        if open(_stamp_path()).read() != _flag_signature():
            return True
    except OSError:
        return True
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return _newest(deps) > os.path.getmtime(LIB)


def build(force=False, verbose=False):
    if not (force or needs_build()):
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build", os.path.basename(LIB))
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src + ".o")
        extra = _extra_flags()
        cmd = [hipcc] + FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stderr)
            raise subprocess.CalledProcessError(res.returncode, cmd)
        return obj, kernel_resources(res.stderr)

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        done = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in done]
    resources = {}
    for src, (_, r) in zip(SOURCES, done):
        for k, v in r.items():
            resources[k] = dict(v, source=src)
    spilled = sorted(k for k, v in resources.items() if v["vgpr_spill"] > 0)
    agpr = sorted(k for k, v in resources.items() if v["agprs"] > 0 and not any(demangle(k).startswith("void " + t + "(") or demangle(k).startswith(t + "(")
                                                                                 for t in AGPR_KERNELS_TESTED))
    if (spilled or agpr) and os.environ.get("P3D_ALLOW_SPILLS"):
        # The escape hatch (another ROCm release, another compiler): P3D_ALLOW_SPILLS=1 builds anyway and SAYS which kernels are not
        # to be trusted; run tests/test_gpu_meshes.py::test_all_queue_capacities and the two tests named in AGPR_KERNELS_TESTED on
        # that build before using it.  tests/test_cpu_abi_and_host.py::test_no_kernel_of_the_library_spills_vgprs fails on it.
        print("[build] P3D_ALLOW_SPILLS: kernels with spilled VGPRs:", ", ".join(demangle(k) for k in spilled) or "none",
              "| with untested AGPR use:", ", ".join(demangle(k) for k in agpr) or "none", file=sys.stderr)
    elif agpr:
        raise RuntimeError("kernels that keep registers in AGPRs and are not in build.py: AGPR_KERNELS_TESTED (see there; "
                           "P3D_ALLOW_SPILLS=1 builds anyway): " + ", ".join(f"{demangle(k)} [{resources[k]['agprs']}]" for k in agpr))
    elif spilled:
        # Round 4: every kernel of this library that spilled VGPRs next to SGPR spills LOST queue entries on the GPU (the
        # generic K = 5..7 kernel, TopKReg<32+, 0>, TopKPairs<64, ., 0>: profiles/r04/spill_miscompile.md), bit-exact
        # against the oracle as soon as the same code fitted its registers.  A spill is therefore a build error here, not a
        # performance note: change the launch bounds / queue of the kernel.
        raise RuntimeError("kernels with VGPR spills (results are not trusted, see profiles/r04/spill_miscompile.md; "
                           "P3D_ALLOW_SPILLS=1 builds anyway): " +
                           ", ".join(f"{demangle(k)} [{resources[k]['vgpr_spill']}]" for k in spilled))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(_stamp_path(), "w") as f:
        f.write(_flag_signature())
    with open(LIB + ".resources.json", "w") as f:
        json.dump(resources, f, indent=0, sort_keys=True)
    return LIB


def kernel_resources(remarks):
    """{mangled kernel name: {vgprs, agprs, sgprs, scratch, sgpr_spill, vgpr_spill, occupancy, lds}} from the compiler's
    -Rpass-analysis=kernel-resource-usage remarks."""
    out = {}
    for block in re.split(r"remark: Function Name: ", remarks)[1:]:
        name = block.split()[0]

        def g(key):
            m = re.search(re.escape(key) + r": (\d+)", block)
            return int(m.group(1)) if m else -1

        out[name] = {"vgprs": g("VGPRs"), "agprs": g("AGPRs"), "sgprs": g("TotalSGPRs"), "scratch": g("ScratchSize [bytes/lane]"),
                     "sgpr_spill": g("SGPRs Spill"), "vgpr_spill": g("VGPRs Spill"), "occupancy": g("Occupancy [waves/SIMD]"),
                     "lds": g("LDS Size [bytes/block]")}
    return out


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("p3d::(anonymous namespace)::", "")
    except OSError:
        return name


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
