"""Stage the reference's PYTHON package and its own unit tests for the GPU box (TEST INFRASTRUCTURE).

`/root/reference` exists only in the build container.  The reference's CUDA-half unit tests are the gate the
survey planned for the drop-in boundary, so `stage()` copies

    /root/reference/pytorch3d/**/*.py      (pure Python; no csrc, no implicitron)
    /root/reference/tests/{common_testing,test_*}.py  for the hot-path test modules listed below
    /root/reference/tests/data/*.png, *.jpg (image fixtures of test_render_points / test_render_meshes / ...)
    /root/reference/docs/tutorials/data/cow_mesh/*   (BASELINE configs[1]: the cow)

into oracle/_ref/reference_py/ -- git-ignored like the rest of oracle/_ref/ (nothing of the reference enters the
history), but NOT gpurun-ignored, so it travels to the GPU box next to oracle/_ref/p3d_ref_cpu.so.  Only
tests/run_reference_suite.py reads it; the product package never does.
"""
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("P3D_REFERENCE_ROOT", "/root/reference")
STAGE = os.path.join(HERE, "_ref", "reference_py")

TEST_MODULES = [
    "common_testing", "test_rasterize_meshes", "test_rasterize_points", "test_compositing",
    "test_interpolate_face_attributes", "test_blending", "test_render_points", "test_render_meshes",
    "test_render_meshes_clipped", "test_rasterize_rectangle_images", "test_texturing", "test_shader",
    "test_rasterizer", "test_render_multigpu", "test_face_areas_normals", "test_meshes",
]


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE, "pytorch3d", "renderer"))


def staged():
    return os.path.isdir(os.path.join(STAGE, "pytorch3d", "renderer")) and os.path.isdir(os.path.join(STAGE, "tests"))


def _copy(src, dst):
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    if not os.path.exists(dst) or os.path.getmtime(src) > os.path.getmtime(dst) or os.path.getsize(src) != os.path.getsize(dst):
        shutil.copy2(src, dst)


def stage(force=False):
    if not have_reference():
        return STAGE if staged() else None
    if force and os.path.isdir(STAGE):
        shutil.rmtree(STAGE)
    pkg = os.path.join(REFERENCE, "pytorch3d")
    for d, dirs, files in os.walk(pkg):
        rel = os.path.relpath(d, pkg)
        top = rel.split(os.sep)[0]
        if top in ("csrc", "implicitron"):
            dirs[:] = []
            continue
        for f in files:
            if f.endswith(".py"):
                _copy(os.path.join(d, f), os.path.join(STAGE, "pytorch3d", rel, f))
    tdir = os.path.join(REFERENCE, "tests")
    _copy(os.path.join(tdir, "__init__.py"), os.path.join(STAGE, "tests", "__init__.py"))
    for m in TEST_MODULES:
        src = os.path.join(tdir, m + ".py")
        if os.path.exists(src):
            _copy(src, os.path.join(STAGE, "tests", m + ".py"))
    ddir = os.path.join(tdir, "data")
    for f in (os.listdir(ddir) if os.path.isdir(ddir) else []):
        if f.endswith((".png", ".jpg")):
            _copy(os.path.join(ddir, f), os.path.join(STAGE, "tests", "data", f))
    for sub in ("missing_usemtl", "missing_files_obj", "obj_mtl_no_image"):  # small OBJ fixtures of test_render_meshes
        for d, _dirs, files in os.walk(os.path.join(ddir, sub)):
            for f in files:
                src = os.path.join(d, f)
                _copy(src, os.path.join(STAGE, "tests", "data", os.path.relpath(src, ddir)))
    cow = os.path.join(REFERENCE, "docs", "tutorials", "data", "cow_mesh")
    for f in (os.listdir(cow) if os.path.isdir(cow) else []):
        _copy(os.path.join(cow, f), os.path.join(STAGE, "docs", "tutorials", "data", "cow_mesh", f))
    return STAGE


if __name__ == "__main__":
    import sys

    print(stage(force="--force" in sys.argv))
