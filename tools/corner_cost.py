#!/usr/bin/env python3
"""Cost of the point-to-line extension next to the surf-only loop.

    python tools/corner_cost.py [--batch 128] [--sensor hdl64] [--keyframes 60]

Registers the same batch with and without edge points and prints launches, device ms per
GN iteration (HIP events around every launch unit, cfg.profile = 1) and the pose RMSE vs truth.
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--sensor", default="hdl64")
    ap.add_argument("--keyframes", type=int, default=60)
    a = ap.parse_args()
    pkg = importlib.import_module("lio-slam_amd")
    synth = importlib.import_module("lio-slam_amd.synth")
    import torch
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    case = synth.make_case(a.sensor, n_keyframes=a.keyframes, n_queries=a.batch, device=dev)
    synth.add_corners(case, a.sensor)
    qs = case["queries"]
    scans = [q["scan"] for q in qs]
    corners = [q["corners"] for q in qs]
    poses0 = np.stack([q["pose_init"] for q in qs])
    truth = np.stack([q["pose_true"] for q in qs])
    print(f"map {len(case['map'])} pts, corner map {len(case['corner_map'])} pts, "
          f"surf/scan {np.mean([len(s) for s in scans]):.0f}, edge/scan {np.mean([len(c) for c in corners]):.0f}")
    for with_c in (0, 1):
        s2m = pkg.ScanToMap(profile=1)
        s2m.set_map(case["map"])
        s2m.set_corner_map(case["corner_map"])
        for rep in range(3):
            s2m.batch_upload(scans)
            if with_c:
                s2m.batch_upload_corners(corners)
            s2m.batch_set_poses(poses0)
            t0 = time.perf_counter()
            s2m.batch_run()
            s2m.batch_sync()
            wall = time.perf_counter() - t0
            poses, res = s2m.batch_results()
        p = s2m.profile()
        ms = [p.launch_ms[i] for i in range(p.n_units)]
        err_t = np.sqrt(np.mean((poses[:, 3:] - truth[:, 3:]) ** 2))
        err_r = np.sqrt(np.mean((poses[:, :3] - truth[:, :3]) ** 2))
        print(f"corners={with_c}: launches {p.n_launches}, first-iteration ms {ms[0]:.3f}, sum ms {sum(ms):.3f}, "
              f"wall ms {wall * 1e3:.2f}, mean iters {np.mean([r.iters for r in res]):.2f}, "
              f"rmse {err_t * 100:.2f} cm / {np.degrees(err_r):.4f} deg")
        s2m.close()


if __name__ == "__main__":
    main()
