"""Record every `pytorch3d._C` hot-path call the REFERENCE's OWN TEST-SUITE makes on CPU tensors
(run in the build container only; /root/reference is read, never written).

    python tests/golden/record_reference_suite.py

The reference's unittest modules for this path (tests/test_rasterize_meshes.py,
tests/test_rasterize_points.py, tests/test_compositing.py, tests/test_blending.py) are run unmodified with `pytorch3d._C`
bound to the reference's own CPU kernels (oracle/_ref/p3d_ref_cpu.so).  A thin recorder around each
operator stores (operator, inputs, outputs, the unittest id that made the call).  The tests that pass
are the ones that compare those outputs with the suite's hand-written golden tensors
(_simple_triangle_raster, _test_perspective_correct, _test_barycentric_clipping, _test_back_face_culling,
test_order_of_ties, _test_coarse_rasterize, the 5x5 / 16x16 point goldens, the 4x4 compositing goldens),
so a replay that reproduces the recorded outputs reproduces those goldens.

Output: tests/golden/ref_suite_calls.npz + ref_suite_calls.json (manifest).  Replayed by
tests/test_cpu_reference_suite_replay.py (oracle) and tests/test_gpu_reference_suite_replay.py (HIP).
"""
import json
import os
import sys
import unittest

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REFERENCE = "/root/reference"

OPS = (
    "rasterize_meshes", "rasterize_meshes_backward", "_rasterize_meshes_naive", "_rasterize_meshes_coarse",
    "rasterize_points", "rasterize_points_backward", "_rasterize_points_naive", "_rasterize_points_coarse",
    "accum_alphacomposite", "accum_alphacomposite_backward", "accum_weightedsumnorm",
    "accum_weightedsumnorm_backward", "accum_weightedsum", "accum_weightedsum_backward",
    "sigmoid_alpha_blend", "sigmoid_alpha_blend_backward",
)
MAX_CALL_BYTES = 1 << 20   # skip calls larger than 1 MiB (the suite's benchmark-sized cases)
MAX_TOTAL_BYTES = 12 << 20

calls = []
current_test = ["?"]
seen = set()
total_bytes = [0]


def _enc(v):
    if isinstance(v, torch.Tensor):
        return ("tensor", v.detach().cpu().contiguous().numpy())
    if isinstance(v, (tuple, list)):
        return ("tuple", [int(x) for x in v])
    if isinstance(v, bool):
        return ("bool", bool(v))
    if isinstance(v, int):
        return ("int", int(v))
    if isinstance(v, float):
        return ("float", float(v))
    raise TypeError(type(v))


def make_recorder(name, fn):
    def rec(*args):
        out = fn(*args)
        if any(isinstance(a, torch.Tensor) and a.is_cuda for a in args):
            return out
        outs = out if isinstance(out, (tuple, list)) else (out,)
        enc_in = [_enc(a) for a in args]
        enc_out = [_enc(o) for o in outs]
        nbytes = sum(v.nbytes for k, v in enc_in + enc_out if k == "tensor")
        key = (name, tuple((k, v.tobytes() if k == "tensor" else repr(v)) for k, v in enc_in))
        h = hash(key)
        if nbytes <= MAX_CALL_BYTES and h not in seen and total_bytes[0] + nbytes <= MAX_TOTAL_BYTES:
            seen.add(h)
            total_bytes[0] += nbytes
            calls.append({"op": name, "test": current_test[0], "in": enc_in, "out": enc_out})
        return out

    return rec


class Result(unittest.TextTestResult):
    def startTest(self, test):
        current_test[0] = test.id()
        super().startTest(test)


def main():
    from oracle import oracle as orc

    ref = orc.ref_module()
    assert ref is not None, "build oracle/_ref first (python oracle/build.py)"
    import types

    mod = types.ModuleType("pytorch3d._C")
    for n in dir(ref):
        if not n.startswith("__"):
            setattr(mod, n, getattr(ref, n))
    for n in OPS:
        setattr(mod, n, make_recorder(n, getattr(ref, n)))
    for n, v in dict(EPS=1e-6, MAX_FLOAT=3.4e38, MAX_INT=2147483647, MAX_UINT=4294967295, MAX_USHORT=65535,
                     PULSAR_MAX_GRAD_SPHERES=128).items():
        setattr(mod, n, v)
    sys.modules["pytorch3d._C"] = mod
    sys.path.insert(0, REFERENCE)
    import pytorch3d

    pytorch3d._C = mod
    os.chdir(REFERENCE)
    names = ["tests.test_rasterize_meshes", "tests.test_rasterize_points", "tests.test_compositing", "tests.test_blending"]
    suite = unittest.defaultTestLoader.loadTestsFromNames(names)
    runner = unittest.TextTestRunner(resultclass=Result, verbosity=0, stream=open(os.devnull, "w"))
    res = runner.run(suite)
    passed = res.testsRun - len(res.failures) - len(res.errors) - len(res.skipped)
    bad = {t.id() for t, _ in res.failures + res.errors}
    kept = [c for c in calls if c["test"] not in bad]
    print(f"reference suite: ran {res.testsRun}, passed {passed}, failed/errored {len(bad)} (CUDA-only tests, no GPU here), "
          f"skipped {len(res.skipped)}; recorded {len(kept)} calls from passing tests ({len(calls) - len(kept)} dropped)")

    arrays = {}
    manifest = []
    for i, c in enumerate(kept):
        ent = {"op": c["op"], "test": c["test"], "in": [], "out": []}
        for side in ("in", "out"):
            for j, (k, v) in enumerate(c[side]):
                if k == "tensor":
                    key = f"c{i}_{side}{j}"
                    arrays[key] = v
                    ent[side].append({"t": "tensor", "key": key})
                else:
                    ent[side].append({"t": k, "v": v})
        manifest.append(ent)
    np.savez_compressed(os.path.join(HERE, "ref_suite_calls.npz"), **arrays)
    with open(os.path.join(HERE, "ref_suite_calls.json"), "w") as f:
        json.dump({"reference": "facebookresearch/pytorch3d v0.7.9, CPU kernels (oracle/_ref)", "calls": manifest}, f,
                  indent=0)
    by_op = {}
    for c in kept:
        by_op[c["op"]] = by_op.get(c["op"], 0) + 1
    print(by_op)
    print("tests that contributed:", sorted({c["test"].split(".")[-1] for c in kept}))


if __name__ == "__main__":
    main()
