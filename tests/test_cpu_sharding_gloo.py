"""CPU suite, part 4: the N>1 path on 2 gloo ranks.

The batch of independent (mesh, camera) jobs is partitioned by cost, every rank rasterizes its own
slice (here with the oracle as the stand-in compute function -- the driver is agnostic to it), local
face ids are rebased, and ONE all_gather assembles the full batch: the result must equal the
un-sharded run bit for bit.  bench.py --gpus N uses the same partition/gather code over RCCL.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _util as U


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem():
    verts, faces = U.hetero_batch(5, seed=4, fmin=60, fmax=400)
    fv = torch.cat([v[f] for v, f in zip(verts, faces)], 0)
    cnt = torch.tensor([f.shape[0] for f in faces])
    first = torch.cumsum(cnt, 0) - cnt
    return fv, first, cnt


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from pytorch3d_amd import sharding

        fv, first, cnt = _problem()
        size, blur, K = (24, 24), 1e-3, 3
        parts = sharding.partition([float(c) for c in cnt], world)
        start, stop = parts[rank]
        lo, hi, f_local, c_local = sharding.shard_packed(first, cnt, start, stop)
        nbr = torch.full((hi - lo,), -1, dtype=torch.int64)
        if stop > start:
            p2f, zbuf, bary, dists = orc.rasterize_meshes_naive(fv[lo:hi], f_local, c_local, nbr, size, blur, K, True,
                                                                True, False)
            p2f = sharding.rebase_indices(p2f, lo)
        else:
            p2f = torch.zeros((0,) + size + (K,), dtype=torch.int64)
            zbuf = torch.zeros((0,) + size + (K,))
        sizes = [b - a for a, b in parts]
        full_idx = sharding.gather_batch(p2f, sizes)
        full_z = sharding.gather_batch(zbuf, sizes)
        to_zero = sharding.gather_batch(zbuf, sizes, dst=0)  # bench.py's variant: only rank 0 receives
        assert (to_zero is None) == (rank != 0)
        if rank == 0:
            assert torch.equal(to_zero, full_z)
        torch.save({"idx": full_idx, "z": full_z, "parts": parts}, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_render_equals_unsharded(tmp_path):
    from oracle import oracle as orc

    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    fv, first, cnt = _problem()
    nbr = torch.full((fv.shape[0],), -1, dtype=torch.int64)
    ref = orc.rasterize_meshes_naive(fv, first, cnt, nbr, (24, 24), 1e-3, 3, True, True, False)
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"))
        assert torch.equal(got["idx"], ref[0]), f"rank {r}: gathered pix_to_face differs from the un-sharded run"
        assert torch.equal(got["z"], ref[1])
        parts = got["parts"]
        assert parts[0][0] == 0 and parts[-1][1] == 5 and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))


def test_partition_properties():
    from pytorch3d_amd import sharding

    for costs, w in (([1.0] * 8, 4), ([5, 1, 1, 1, 1, 1], 2), ([1, 2, 3], 8), ([], 2), ([10, 1], 2)):
        parts = sharding.partition(costs, w)
        assert len(parts) == w
        assert parts[0][0] == 0 and parts[-1][1] == len(costs)
        assert all(a[1] == b[0] and a[0] <= a[1] for a, b in zip(parts, parts[1:]))
    parts = sharding.partition([1.0] * 8, 4)
    assert [b - a for a, b in parts] == [2, 2, 2, 2]
    heavy = sharding.partition([8, 1, 1, 1, 1, 1, 1, 1, 1], 2)
    assert heavy[0] == (0, 1)


def test_shard_and_rebase():
    from pytorch3d_amd import sharding

    first = torch.tensor([0, 4, 9, 11])
    cnt = torch.tensor([4, 5, 2, 6])
    lo, hi, f, c = sharding.shard_packed(first, cnt, 1, 3)
    assert (lo, hi) == (4, 11) and f.tolist() == [0, 5] and c.tolist() == [5, 2]
    idx = torch.tensor([-1, 0, 6])
    assert sharding.rebase_indices(idx, lo).tolist() == [-1, 4, 10]
    lo, hi, f, c = sharding.shard_packed(first, cnt, 2, 2)
    assert (lo, hi) == (0, 0) and f.numel() == 0


def test_bench_dry_run_maps_ranks_to_devices_without_a_gpu():
    """`python bench.py --gpus 4 --dry-run` (bench.py: dry_run): the launcher, the RANK / LOCAL_RANK -> device mapping and the gloo
    rendezvous of an N-GPU run, with no RCCL and no kernel -- runs in this container, where no device is visible: the line must
    list every rank and say that its device does not exist."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--dry-run"], capture_output=True, text=True, timeout=240, cwd=root, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-1000:]
    j = json.loads(lines[0])
    assert j["dry_run"] is True and j["n_gpus"] == 4
    assert [m["rank"] for m in j["mapping"]] == [0, 1, 2, 3] and [m["device"] for m in j["mapping"]] == [f"cuda:{i}" for i in range(4)]
    visible = j["mapping"][0]["devices_visible"]
    assert len([p for p in j["problems"] if "has no device" in p]) == max(0, 4 - visible) and j["ok"] == (not j["problems"])
