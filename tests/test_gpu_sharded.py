"""Two ranks sharing the one GPU of the test box (gloo carries the collective;
on a multi-GPU node bench.py uses RCCL): the sharded-map path must reproduce the
unsharded registration."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, deterministic, mode="map", groups=2):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    pkg = importlib.import_module("lio-slam_amd")
    mg = importlib.import_module("lio-slam_amd.multigpu")
    synth = importlib.import_module("lio-slam_amd.synth")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = synth.make_case("vlp16", n_keyframes=6, seed=11, device="cpu", n_queries=3)
        scans = [qq["scan"] for qq in case["queries"]] + [case["queries"][0]["scan"][:25]]
        poses0 = np.stack([qq["pose_init"] for qq in case["queries"]] + [case["queries"][0]["pose_init"]])
        runner = mg.ShardedRunner(pkg, case["map"], rank, world, dist, torch, mode=mode, groups=groups,
                                  deterministic=deterministic, device_id=0)
        runner.upload(scans)
        runner.set_poses(poses0)
        n_it = runner.run()
        poses, res = runner.results()
        out = {"poses": poses, "iters": [r.iters for r in res], "status": [r.status for r in res],
               "deg": [r.is_degenerate for r in res], "n_it": n_it, "n_shard": len(runner.idx), "n_map": len(case["map"])}
        if rank == 0:
            ref = pkg.ScanToMap(device_id=0)
            ref.set_map(case["map"])
            ref.batch_upload(scans); ref.batch_set_poses(poses0); ref.batch_run()
            rp, rr = ref.batch_results()
            out["ref_poses"] = rp
            out["ref_iters"] = [r.iters for r in rr]
            out["ref_status"] = [r.status for r in rr]
            ref.close()
        runner.close()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("deterministic,mode,groups", [(False, "map", 2), (True, "map", 1), (False, "scan", 2), (True, "scan", 3)])
def test_two_rank_sharded_map_matches_unsharded(deterministic, mode, groups):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, deterministic, mode, groups)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a, b = res[0], res[1]
    np.testing.assert_array_equal(a["poses"], b["poses"])            # every rank solves the same sums
    assert a["iters"] == b["iters"] == a["ref_iters"]
    assert a["status"] == b["status"] == a["ref_status"]
    if mode == "map":
        assert a["n_shard"] < a["n_map"] and b["n_shard"] < b["n_map"]
    np.testing.assert_allclose(a["poses"][:, 3:], a["ref_poses"][:, 3:], atol=1e-5)
    np.testing.assert_allclose(a["poses"][:, :3], a["ref_poses"][:, :3], atol=1e-6)
    assert a["status"][-1] == 1                                       # the too-small scan is skipped (MO:1844)
