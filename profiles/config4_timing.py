#!/usr/bin/env python
"""BASELINE configs[3] alone (1M points, 512^2, K = 10, rasterizer + alpha compositor, forward + backward): per-kernel times
through bench.py's own `other_configs` leg, one JSON line.  `python profiles/config4_timing.py`"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pytorch3d_amd import _lib  # noqa: E402

out = bench.other_configs(_lib.load(), _lib, torch.device("cuda:0"))["config4_points_1m_512_k10_fwd_bwd"]
print(json.dumps({"wall_ms": out.get("wall_ms"), "kernels_ms": out.get("kernels_ms"), "error": out.get("error")}))
