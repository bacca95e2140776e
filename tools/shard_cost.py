#!/usr/bin/env python3
"""Per-rank cost of one GN-iteration launch under the two N>1 partitions, measured on ONE GPU
(no collective): weak scaling = 512*N scans in the job, rank r does its share.
   python tools/shard_cost.py /tmp/c512.npz"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("lio-slam_amd")
mg = importlib.import_module("lio-slam_amd.multigpu")
z = np.load(sys.argv[1])
lens = z["lens"][:512]
offs = np.concatenate([[0], np.cumsum(lens)])
cat = z["scans"]                  # (an NpzFile re-reads the member on every access)
scans1 = [np.ascontiguousarray(cat[offs[i]:offs[i + 1]]) for i in range(len(lens))]
map_xyz, poses1 = z["map"], z["poses0"][:512]


HALO = int(os.environ.get("SHARD_HALO", "16"))


def one(mode, world, rank, balance="load"):
    s2m = pkg.ScanToMap(profile=1, max_iters=2)
    if mode == "map":
        synth = importlib.import_module("lio-slam_amd.synth")
        load = np.concatenate([mg.transform_f32(synth.pose_matrix(poses1[i])[:3].astype(np.float32).reshape(12), scans1[i][::16])
                               for i in range(0, len(scans1), 4)]) if balance == "load" else None
        plan = mg.plan_shards(map_xyz, world, load_xyz=load)
        idx = mg.shard_points(map_xyz, plan, rank, HALO)
        s2m.set_map(np.ascontiguousarray(map_xyz[idx]))
        s2m.set_global_grid([float(v) for v in plan["origin"]], [int(v) for v in plan["dims"]])
        s2m.set_shard_plan(plan["axis"], rank, [int(v) for v in plan["bounds"]], HALO)
    else:
        s2m.set_map(map_xyz)
        s2m.set_scan_shard(rank, world)
    s2m.batch_upload(scans1 * world)
    s2m.batch_set_poses(np.tile(poses1, (world, 1)))
    sums = torch.zeros((len(scans1) * world, 32), dtype=torch.float64, device="cuda")
    ms = []
    for rep in range(3):
        s2m.batch_begin()
        s2m.batch_iter_partial(sums.data_ptr())
        s2m.batch_sync()
        s2m.batch_results(False)
        ms.append(s2m.profile().launch_ms[0])
    s2m.close()
    return min(ms)


base = one("scan", 1, 0)
print(f"N=1: {base:.3f} ms per launch (512 scans)")
for mode in ("map", "map/map-point-balanced", "scan"):
    for world in (2, 4, 8):
        t = [one(mode.split("/")[0], world, r, "map" if "/" in mode else "load") for r in range(world)]
        print(f"{mode:4s} N={world}: slowest rank {max(t):.3f} ms for {512 * world} scans -> kernel-only weak-scaling efficiency {base / max(t):.2f}")
