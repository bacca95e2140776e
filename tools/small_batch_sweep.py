#!/usr/bin/env python3
"""Small batches (1..8 scans of the bench case, one workgroup per compute unit at most): the launch loop (eager and
hipGraph) against the one-launch loop (cfg.pipeline = 4).  batch_upload + set_poses + run + results per step, host
buffers.   python tools/small_batch_sweep.py [case.npz]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("lio-slam_amd")
synth = importlib.import_module("lio-slam_amd.synth")
if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
    z = np.load(sys.argv[1])
    lens = z["lens"][:32]
    offs = np.concatenate([[0], np.cumsum(lens)])
    cat = z["scans"]
    qs = [{"scan": np.ascontiguousarray(cat[offs[i]:offs[i + 1]]), "pose_init": z["poses0"][i]} for i in range(32)]
    map_xyz = z["map"]
else:
    case = synth.make_case("hdl64", n_keyframes=60, n_queries=32)
    qs, map_xyz = case["queries"], case["map"]
print(f"# batch_upload + set_poses + run + results, hdl64 scans N_s ~ {np.mean([len(q['scan']) for q in qs]):.0f} vs N_m = {len(map_xyz)}; ms per batch")
for nb in (1, 2, 4, 8, 9):
    row = []
    for name, cfg in [("eager", dict(pipeline=1)), ("graph 6", dict(pipeline=1, use_graph=1, graph_iters=6)), ("one launch", dict(pipeline=4))]:
        s2m = pkg.ScanToMap(**cfg)
        s2m.set_map(map_xyz)
        groups = [qs[i:i + nb] for i in range(0, len(qs) - nb + 1, nb)]
        def step(g):
            s2m.batch_upload([q["scan"] for q in g]); s2m.batch_set_poses(np.stack([q["pose_init"] for q in g])); s2m.batch_run()
            return s2m.batch_results(with_results=False)
        for g in groups[:2]:
            step(g)
        t = time.perf_counter()
        reps = 4
        for _ in range(reps):
            for g in groups:
                step(g)
        dt = 1e3 * (time.perf_counter() - t) / (reps * len(groups))
        row.append(f"{name} {dt:.3f}" + (f" (pipeline {s2m.profile().pipeline})" if name == "one launch" else ""))
        s2m.close()
    print(f"{nb} scans: " + ", ".join(row))
