import importlib, sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
pkg = importlib.import_module("lio-slam_amd"); mg = importlib.import_module("lio-slam_amd.multigpu")
z = np.load(sys.argv[1]); lens = z["lens"][:512]; offs = np.concatenate([[0], np.cumsum(lens)]); cat = z["scans"]
scans = [np.ascontiguousarray(cat[offs[i]:offs[i+1]]) for i in range(512)]; map_xyz = z["map"]; poses = z["poses0"][:512]
HALO = int(os.environ.get("SHARD_HALO", "16"))
WORLDS = [int(v) for v in os.environ.get("WORLDS", "2,8").split(",")]
RANKS = os.environ.get("RANKS")
for world in WORLDS:
    plan = mg.plan_shards(map_xyz, world)
    print("axis", plan["axis"], "bounds", plan["bounds"], "dims", plan["dims"])
    for rank in ([int(v) for v in RANKS.split(",")] if RANKS else range(world)):
        s2m = pkg.ScanToMap(profile=1, max_iters=2, xcd_remap=int(os.environ.get("XCD", "1")), sort_batch=int(os.environ.get("SORTB", "1")))
        idx = mg.shard_points(map_xyz, plan, rank, HALO)
        s2m.set_map(np.ascontiguousarray(map_xyz[idx])); s2m.set_global_grid([float(v) for v in plan["origin"]], [int(v) for v in plan["dims"]])
        s2m.set_shard_plan(plan["axis"], rank, [int(v) for v in plan["bounds"]], HALO)
        rep = int(os.environ.get("WEAK", "0")) and world or 1
        s2m.batch_upload(scans * rep); s2m.batch_set_poses(np.tile(poses, (rep, 1)))
        sums = torch.zeros((512 * rep, 32), dtype=torch.float64, device="cuda")
        ms=[]
        for rep in range(3):
            s2m.batch_begin(); s2m.batch_iter_partial(sums.data_ptr()); s2m.batch_sync(); s2m.batch_results(False); ms.append(s2m.profile().launch_ms[0])
        nc = float(sums[:, 27].sum().item())
        print(f"world {world} rank {rank}: {min(ms):.3f} ms, correspondences {nc:.0f}, shard pts {len(idx)}")
        s2m.close()
