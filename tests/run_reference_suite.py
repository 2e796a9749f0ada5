#!/usr/bin/env python
"""Run the REFERENCE's own unit tests against pytorch3d_amd through the `pytorch3d._C` shim (test infrastructure).

    python tests/run_reference_suite.py --out gpurun_out/ref_suite.json [module ...]

The reference's pure-Python package and its test modules are taken from oracle/_ref/reference_py/ (staged by
oracle/stage_reference.py in the build container, git-ignored, shipped to the GPU box by gpurun).  `pytorch3d._C` is
pytorch3d_amd's operator surface for CUDA (= HIP) tensors.  The reference's tests also call `_C` with CPU tensors
(their "cpu vs cuda" comparisons): those calls are routed to the reference's OWN CPU kernels, compiled from its
sources into oracle/_ref/p3d_ref_cpu.so -- so a "cpu vs cuda" test of the reference compares the reference's CPU
implementation with our HIP kernels, which is exactly the parity statement wanted.  The product package has no such
routing: pytorch3d_amd._C refuses CPU tensors.

Output: one JSON object {module: {test id: {"outcome": pass|fail|error|skip, "msg": ...}}} plus a summary on stdout.
Packages the image lacks and the tests import but the hot path never uses (iopath, imageio, OpenGL) are stubbed.
"""
import argparse
import importlib
import importlib.util
import io
import json
import os
import sys
import time
import types
import unittest
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.path.join(ROOT, "oracle", "_ref", "reference_py")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "p3d_ref_cpu.so")

DEFAULT_MODULES = ["test_rasterize_meshes", "test_rasterize_points", "test_compositing",
                   "test_interpolate_face_attributes", "test_blending", "test_render_points", "test_render_meshes",
                   "test_rasterize_rectangle_images", "test_texturing", "test_shader", "test_render_meshes_clipped",
                   "test_rasterizer"]


def _stub_missing_packages():
    """iopath / imageio are not in this image.  pytorch3d.io only needs PathManager.open / exists / get_local_path on
    local files; imageio is only used by debug dumps."""
    try:
        import iopath  # noqa: F401
    except ImportError:
        iop = types.ModuleType("iopath")
        common = types.ModuleType("iopath.common")
        file_io = types.ModuleType("iopath.common.file_io")

        class PathManager:
            def open(self, path, mode="r", **kw):
                return open(path, mode, **{k: v for k, v in kw.items() if k in ("buffering", "encoding", "errors", "newline")})

            def exists(self, path):
                return os.path.exists(path)

            def isfile(self, path):
                return os.path.isfile(path)

            def get_local_path(self, path, **kw):
                return str(path)

            def mkdirs(self, path):
                os.makedirs(path, exist_ok=True)

        file_io.PathManager = PathManager
        file_io.g_pathmgr = PathManager()
        common.file_io = file_io
        iop.common = common
        sys.modules.update({"iopath": iop, "iopath.common": common, "iopath.common.file_io": file_io})
    try:
        import imageio  # noqa: F401
    except ImportError:
        im = types.ModuleType("imageio")
        im.imwrite = lambda *a, **k: None
        im.imread = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("imageio stub"))
        sys.modules["imageio"] = im
    # tests/test_rasterizer.py imports pure-Python helpers (_check_cameras, _parse_and_verify_image_size, ...) from
    # pytorch3d.renderer.opengl.rasterizer_opengl, whose module body imports pyopengl / pycuda.  The EGL rasterizer itself is
    # outside the path and its tests are skipped (PYTORCH3D_NO_TEST_OPENGL=1): any attribute of the stubs is an inert class.
    try:
        import OpenGL.EGL  # noqa: F401
    except Exception:
        class _InertMeta(type):
            def __getattr__(cls, name):
                if name.startswith("__"):
                    raise AttributeError(name)
                return cls

        class _Inert(Exception, metaclass=_InertMeta):
            def __init__(self, *a, **k):
                pass

            def __call__(self, *a, **k):
                return _Inert()

            def __getattr__(self, name):
                return _Inert

        class _InertModule(types.ModuleType):
            def __getattr__(self, name):
                if name.startswith("__"):
                    raise AttributeError(name)
                return _Inert

        for name in ("OpenGL", "OpenGL.GL", "OpenGL.EGL", "OpenGL._opaque", "OpenGL.raw", "OpenGL.raw.EGL", "OpenGL.raw.EGL._errors",
                     "pycuda", "pycuda.gl", "pycuda.driver"):
            m = _InertModule(name)
            m.__path__ = []
            sys.modules[name] = m
            if "." in name:  # `import a.b as c` binds getattr(a, "b")
                parent, child = name.rsplit(".", 1)
                setattr(sys.modules[parent], child, m)


def _device_routed_C(hip_from_reference=None):
    """`pytorch3d._C` for the reference's tests: HIP tensors -> pytorch3d_amd, CPU tensors -> the reference's own CPU
    build (oracle/_ref/p3d_ref_cpu.so).  hip_from_reference ("fma" | "nofma"): HIP tensors go to the reference's OWN device
    kernels instead (oracle/_ref/p3d_ref_hip[_nofma].so, the hipified .cu files: oracle/build_ref_hip.py) wherever that
    build has the operator -- the control run that says what the reference's tests do on the reference's code on this GPU."""
    import torch

    import pytorch3d_amd.shim as shim

    mod = shim.make_module()
    mod.__p3d_amd__ = True
    ref = None
    if os.path.exists(REF_SO):
        spec = importlib.util.spec_from_file_location("p3d_ref_cpu", REF_SO)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    counts = {"hip": {}, "ref_cpu": {}}

    def route(name, ours):
        theirs = getattr(ref, name, None) if ref is not None else None

        def call(*args, **kwargs):
            tensors = [a for a in list(args) + list(kwargs.values()) if isinstance(a, torch.Tensor)]
            on_gpu = any(t.is_cuda for t in tensors)
            if not on_gpu and theirs is not None:
                counts["ref_cpu"][name] = counts["ref_cpu"].get(name, 0) + 1
                return theirs(*args, **kwargs)
            counts["hip"][name] = counts["hip"].get(name, 0) + 1
            return ours(*args, **kwargs)

        call.__name__ = name
        return call

    from pytorch3d_amd import _C as ours_C

    ref_hip = None
    if hip_from_reference:
        so = os.path.join(ROOT, "oracle", "_ref", "p3d_ref_hip_nofma.so" if hip_from_reference == "nofma" else "p3d_ref_hip.so")
        if not os.path.exists(so):
            raise SystemExit(f"{so}: the reference's device build is missing (oracle/build_ref_hip.py, build container)")
        spec = importlib.util.spec_from_file_location(os.path.basename(so)[:-3], so)
        ref_hip = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref_hip)
        counts["ref_hip"] = {}

    def route_hip(name, ours):
        theirs = getattr(ref_hip, name, None) if ref_hip is not None else None
        if theirs is None:
            return ours

        def call(*args, **kwargs):
            counts["ref_hip"][name] = counts["ref_hip"].get(name, 0) + 1
            return theirs(*args, **kwargs)

        return call

    for name in ours_C.HOT_PATH_EXPORTS:
        setattr(mod, name, route(name, route_hip(name, getattr(ours_C, name))))

    # The four small operators above the boundary (face areas / normals, packed <-> padded; pytorch3d_amd/_aux_ops.py):
    # HIP tensors -> our torch formulations, CPU tensors -> the reference's CPU kernels, like everything else.
    from pytorch3d_amd import _aux_ops

    for name in ("face_areas_normals_forward", "face_areas_normals_backward", "packed_to_padded", "padded_to_packed"):
        setattr(mod, name, route(name, getattr(_aux_ops, name)))
    return mod, counts


class _Collector(unittest.TextTestResult):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.records = {}
        self._t0 = {}

    def startTest(self, test):
        super().startTest(test)
        self._t0[test.id()] = time.time()

    def _rec(self, test, outcome, msg=""):
        self.records[test.id()] = {"outcome": outcome, "msg": msg[-1500:], "s": round(time.time() - self._t0.get(test.id(), time.time()), 2)}

    def addSuccess(self, test):
        super().addSuccess(test)
        self._rec(test, "pass")

    def addFailure(self, test, err):
        super().addFailure(test, err)
        self._rec(test, "fail", self._exc_info_to_string(err, test))

    def addError(self, test, err):
        super().addError(test, err)
        self._rec(test, "error", self._exc_info_to_string(err, test))

    def addSkip(self, test, reason):
        super().addSkip(test, reason)
        self._rec(test, "skip", reason)

    def addExpectedFailure(self, test, err):
        super().addExpectedFailure(test, err)
        self._rec(test, "pass", "expected failure")

    def addUnexpectedSuccess(self, test):
        super().addUnexpectedSuccess(test)
        self._rec(test, "fail", "unexpected success")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("modules", nargs="*", default=DEFAULT_MODULES)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ref_suite.json"))
    ap.add_argument("--stage", default=STAGE)
    ap.add_argument("-k", dest="pattern", default=None, help="only tests whose id contains this substring")
    ap.add_argument("--patch-python", action="store_true",
                    help="also replace the reference's torch formulations around the operators (clip_faces, softmax_rgb_blend, "
                         "phong_shading, TexturesUV.sample_textures, the face gather ...) with the fused HIP versions "
                         "(pytorch3d_amd.shim.patch_reference_python)")
    ap.add_argument("--hip-from-reference", choices=["fma", "nofma"], default=None,
                    help="control run: HIP tensors go to the reference's own device kernels (oracle/_ref/p3d_ref_hip[_nofma].so)")
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(args.stage, "pytorch3d")):
        raise SystemExit(f"{args.stage}: the reference is not staged (run oracle/stage_reference.py in the build container)")
    os.environ.setdefault("PYTORCH3D_NO_TEST_OPENGL", "1")
    for p in (ROOT, args.stage):
        if p not in sys.path:
            sys.path.insert(0, p)
    _stub_missing_packages()
    warnings.filterwarnings("ignore")
    mod, counts = _device_routed_C(args.hip_from_reference)
    sys.modules["pytorch3d._C"] = mod
    import pytorch3d

    pytorch3d._C = mod
    if args.patch_python:
        import pytorch3d_amd.shim as shim

        shim.patch_reference_python()
    try:  # the EGL rasterizer is outside the path; the tests only need the name to exist at import time
        import pytorch3d.renderer.opengl as ogl

        if not hasattr(ogl, "MeshRasterizerOpenGL"):
            ogl.MeshRasterizerOpenGL = type("MeshRasterizerOpenGL", (), {})
    except Exception:
        pass

    report = {}
    totals = {"pass": 0, "fail": 0, "error": 0, "skip": 0}
    for m in args.modules:
        try:
            tm = importlib.import_module("tests." + m)
        except Exception as e:
            report[m] = {"__import__": {"outcome": "error", "msg": repr(e)}}
            totals["error"] += 1
            print(f"[{m}] import failed: {e!r}", flush=True)
            continue
        suite = unittest.defaultTestLoader.loadTestsFromModule(tm)
        if args.pattern:
            keep = unittest.TestSuite()

            def walk(s):
                for t in s:
                    if isinstance(t, unittest.TestSuite):
                        walk(t)
                    elif args.pattern in t.id():
                        keep.addTest(t)

            walk(suite)
            suite = keep
        stream = io.StringIO()
        res = unittest.TextTestRunner(stream=stream, resultclass=_Collector, verbosity=0).run(suite)
        report[m] = res.records
        c = {"pass": 0, "fail": 0, "error": 0, "skip": 0}
        for r in res.records.values():
            c[r["outcome"]] += 1
            totals[r["outcome"]] += 1
        print(f"[{m}] {c}", flush=True)
        for tid, r in res.records.items():
            if r["outcome"] in ("fail", "error"):
                last = r["msg"].strip().splitlines()[-1] if r["msg"].strip() else ""
                print(f"    {r['outcome'].upper():5s} {tid.split('.', 2)[-1]}: {last[:200]}", flush=True)
    report["__calls__"] = counts
    if args.patch_python:
        import pytorch3d_amd.shim as shim

        report["__patched_calls__"] = {k: {"fused": v[0], "reference_fallback": v[1]} for k, v in sorted(shim.PATCH_CALLS.items())}
        print("patched python calls (fused / fallback):", report["__patched_calls__"], flush=True)
    report["__totals__"] = totals
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    print("TOTAL", totals, "| operator calls", {k: sum(v.values()) for k, v in counts.items()}, flush=True)


if __name__ == "__main__":
    main()
