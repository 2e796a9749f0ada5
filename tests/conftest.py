import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (*.so are git-ignored): build the HIP library (hipcc cross-compiles
    without a GPU; no-op when up to date) so that the ABI / loader tests and the GPU tests find it."""
    try:
        from pytorch3d_amd import build as hip_build

        if hip_build.needs_build():
            hip_build.build()
    except Exception as e:  # no hipcc on this machine: the tests that need the library will say so
        print(f"[conftest] could not build libp3d_amd.so: {e!r}", file=sys.stderr)


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
