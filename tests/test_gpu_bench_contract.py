"""bench.py itself: the single-rank line and the multi-rank code path (two ranks over gloo sharing cuda:0 -- the driver's
N > 1 launch with `P3D_BENCH_TEST_BACKEND=gloo` standing in for RCCL), in both modes: weak scaling (every rank its own
batch) and `--jobs` (BASELINE configs[4]: a fixed set of sub-batches dealt to the ranks, final gather timed separately)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None, timeout=240):
    e = dict(os.environ)
    e.update(env or {})
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def _check_common(j, n_gpus, steps):
    assert j["metric"].startswith("rasterized Mpix/s") and j["unit"] == "Mpix/s" and j["dtype"] == "f32"
    assert j["n_gpus"] == n_gpus and j["steps"] == steps and j["higher_is_better"] is True
    # no published number exists for the metric: null, or (N = 1) the ratio to the reference's own device kernels timed in the run
    assert j["vs_baseline"] is None or (j["vs_baseline"] > 0 and j["vs_reference_device"]["value"] > 0)
    assert j["value"] > 0 and j["ms_per_step"] > 0 and "workload" in j["config"]
    assert j["prewarm_s"] >= 0 and j["prewarm_steps"] >= 0  # untimed steps ahead of the warm-up: disclosed in the line
    r = j["roofline"]
    assert r["bound"] in ("hbm", "valu") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert (r["bound"] == "valu") == (r["valu"] is not None and r["valu"]["frac"] > r["frac"])  # whichever fraction is higher
    for k, v in r["per_kernel"].items():
        assert 0 < v["frac_of_peak_compulsory"] < 1, (k, v)


def test_single_rank_line_small():
    j = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--batch", "4", "--image-size", "128",
              "--no-cpu-baseline"])
    _check_common(j, 1, 3)
    assert j["scaling"] == "weak" and "other_configs" in j
    assert j["without_prewarm"]["value"] > 0 and j["without_prewarm"]["steps"] == 3  # the driver's protocol to the letter, beside the headline
    assert j["roofline"]["valu"] is None  # the committed counters belong to the 64 x 512^2 workload, not to this small one
    rd = j["vs_reference_device"]  # the reference's own device kernels on the same batch, same GPU (or why not)
    assert (rd.get("value") and rd["forward_ms"] > 0 and rd["backward_ms"] > 0 and j["vs_baseline"] == pytest.approx(j["value"] / rd["value"])) \
        or rd.get("reason") or rd.get("error"), rd
    oc = j["other_configs"]
    assert "wall_ms" in oc["config2_cow_256_k8_fwd"], oc
    hg = oc["config2_cow_256_k8_fwd"]["hip_graph"]  # round 5: the same operator call replayed from a HIP graph
    assert hg.get("identical_outputs") is True and hg["graph_replay_ms"] > 0 and hg["eager_C_call_ms"] > 0, hg
    assert "wall_ms" in oc["config4_points_1m_512_k10_fwd_bwd"], oc
    # the unmodified reference MeshRasterizer through the shim, both modes (or a reason where the reference is not staged)
    dr = j["dropin"]
    for mode in ("c_only", "patched"):
        assert mode in dr and ("ms_per_step" in dr[mode] or dr[mode].get("reason")), dr
    if "ms_per_step" in dr["patched"]:
        assert dr["patched"]["grad_finite"] and dr["patched"]["patched_calls"]["MeshRasterizer.forward"][0] > 0, dr
    # the lighter batch of rounds 1-3 rides along as an extra key; the headline is SURVEY 8(d) config 3 as written
    assert j["workload_torus_div_1.5"]["covered_pixel_fraction"] < j["config"]["covered_pixel_fraction"]
    assert "unscaled" in j["config"]["workload"]
    c4 = oc["config4_points_1m_512_k10_fwd_bwd"]
    assert c4["algorithmic_bytes"] > 2.0e8 and set(c4["per_kernel"]) == {"points_fine", "points_backward", "alpha_composite_fwd", "alpha_composite_bwd"}


@pytest.mark.parametrize("mode", ["weak", "jobs"])
def test_two_ranks_over_gloo_on_one_gpu(mode):
    port = 29571 if mode == "weak" else 29572
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--warmup", "1", "--batch", "4", "--image-size", "128"]
    cmd += ["--steps", "2"] if mode == "weak" else ["--jobs", "16"]
    j = _run(cmd, {"P3D_BENCH_TEST_BACKEND": "gloo"})
    if mode == "weak":
        _check_common(j, 2, 2)
        assert j["scaling"] == "weak" and j["config"]["global_batch"] == 8
    else:
        _check_common(j, 2, 2)  # 16 jobs = 4 sub-batches of 4, two per rank
        assert j["scaling"] == "strong" and j["config"]["global_batch"] == 16
    assert j["gather_ms"] > 0
    assert "cpu_baseline" not in j  # rank 0 at N = 1 only
    # every rank's own clock rides in the line (the first real 8-GPU run must yield the efficiency table in one shot)
    assert j["gather"]["ok"] and j["gather"]["backend"] == "gloo" and len(j["per_rank"]) == 2
    assert all(p["seconds"] > 0 and p["mesh_fine_ms"] > 0 and p["gather_ms"] > 0 for p in j["per_rank"])
    assert abs(max(p["seconds"] for p in j["per_rank"]) - j["ms_per_step"] * j["steps"] * 1e-3) < 1e-6


def test_gpus_flag_spawns_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it (the form the driver uses for N = 1) must run TWO ranks and
    print one line with n_gpus = 2 -- not silently one rank (VERDICT round 2, row e)."""
    env = {"P3D_BENCH_TEST_BACKEND": "gloo"}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        assert k not in os.environ, f"{k} is set in the test environment"
    j = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--image-size", "128"], env)
    _check_common(j, 2, 2)
    assert j["scaling"] == "weak" and j["config"]["global_batch"] == 8 and j["gather_ms"] > 0


def test_rccl_failure_falls_back_to_gloo_and_says_so():
    """An RCCL that cannot be brought up (IPC mode, topology; injected here with P3D_BENCH_FAIL_NCCL -- two ranks on one device
    make the real RCCL hang instead of fail): the run must still produce its line -- barriers, timing and the gather over gloo
    -- and say what happened in `gather.backend_note` instead of dying in init_process_group."""
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        assert k not in os.environ, f"{k} is set in the test environment"
    j = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--image-size", "128"],
             {"P3D_BENCH_SHARED_GPU": "1", "P3D_BENCH_FAIL_NCCL": "1"}, timeout=240)
    _check_common(j, 2, 2)
    g = j["gather"]
    assert g["backend"] == "gloo" and "nccl failed" in g["backend_note"] and g["ok"], g
    assert len(j["per_rank"]) == 2 and j["gather_ms"] > 0


def _torchrun(n, port, extra, timeout=200):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", str(n), "--warmup", "1", "--batch", "2", "--image-size", "64",
           "--prewarm-s", "0", "--gather-checksum", "--no-other-configs"] + extra
    return _run(cmd, {"P3D_BENCH_TEST_BACKEND": "gloo", "OMP_NUM_THREADS": "1"}, timeout=timeout)


def test_eight_ranks_and_an_uneven_deal_gather_what_one_rank_renders():
    """The N > 1 path as far as one GPU allows: EIGHT ranks over gloo sharing cuda:0, weak mode (rank r
    renders generator seed r) and `--jobs` (8 sub-batches dealt to 8 ranks, and to THREE: 3 + 3 + 2), against ONE rank running the
    same 8 sub-batches: the depth images gathered on rank 0 are the same bytes, in job order, in all four runs; every rank's clock
    and kernel time ride in the line."""
    import time

    t0 = time.time()
    single = _run([sys.executable, "bench.py", "--gpus", "1", "--warmup", "1", "--batch", "2", "--image-size", "64", "--jobs", "16",
                   "--prewarm-s", "0", "--gather-checksum", "--no-other-configs", "--no-cpu-baseline", "--no-dropin", "--no-reference-device"])
    ref = single["gather"]["gathered"]
    assert ref["shape"] == [16, 64, 64] and single["config"]["global_batch"] == 16 and single["steps"] == 8
    runs = {"weak x8": _torchrun(8, 29581, ["--steps", "2"]), "jobs x8": _torchrun(8, 29582, ["--jobs", "16"]),
            "jobs x3 (3 + 3 + 2)": _torchrun(3, 29583, ["--jobs", "16"])}
    for name, j in runs.items():
        n = j["n_gpus"]
        got = j["gather"]["gathered"]
        assert j["gather"]["ok"] and (got["sha256"], got["shape"]) == (ref["sha256"], ref["shape"]), (name, j["gather"], ref)
        assert len(j["per_rank"]) == n and all(p["mesh_fine_ms"] > 0 and p["mesh_backward_ms"] > 0 and p["seconds"] > 0 for p in j["per_rank"]), name
        assert j["config"]["global_batch"] == 16 and j["value"] > 0
    assert runs["jobs x3 (3 + 3 + 2)"]["steps"] == 3 and runs["jobs x8"]["steps"] == 1 and runs["weak x8"]["scaling"] == "weak"
    assert time.time() - t0 < 240, "the multi-rank plumbing test is meant to stay short"


def test_dry_run_of_an_eight_gpu_launch_maps_ranks_to_devices():
    """`python bench.py --gpus 8 --dry-run`: the launcher, WORLD_SIZE / RANK / LOCAL_RANK -> device mapping and the rendezvous,
    without RCCL and without a kernel.  On this one-GPU box the line must say which devices are missing."""
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        assert k not in os.environ, f"{k} is set in the test environment"
    j = _run([sys.executable, "bench.py", "--gpus", "8", "--dry-run"], {"OMP_NUM_THREADS": "1"}, timeout=120)
    assert j["dry_run"] is True and j["n_gpus"] == 8 and len(j["mapping"]) == 8
    assert [m["rank"] for m in j["mapping"]] == list(range(8)) and [m["local_rank"] for m in j["mapping"]] == list(range(8))
    assert [m["device"] for m in j["mapping"]] == [f"cuda:{i}" for i in range(8)]
    visible = j["mapping"][0]["devices_visible"]
    missing = [p for p in j["problems"] if "has no device" in p]
    assert len(missing) == max(0, 8 - visible) and j["ok"] == (not j["problems"])
