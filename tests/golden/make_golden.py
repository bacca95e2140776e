#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

The reference (JiLiBIT/LIO-SLAM) holds NO fixtures, golden vectors or tests for
the scan-to-map path and cannot be built or run here, so
these vectors are NOT outputs of the reference: they are inputs from the
synthetic generator (lio-slam_amd/synth.py) and expected outputs of the CPU
oracle (oracle/lio_oracle.c, portable -O2 build).  They pin the oracle against
regressions and give the GPU tests a committed expectation ("parity unpinned"
with respect to a reference binary remains true).

    python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, DeskewConfig  # noqa: E402

synth = importlib.import_module("lio-slam_amd.synth")


def main():
    orc = Oracle(os.path.join(ROOT, "oracle", "liblio_oracle.so"))
    # ---- registration: VLP-16 vs 5 keyframes, street + corridor (degenerate)
    for name, kind, seed in (("s2m_street", "street", 21), ("s2m_corridor", "corridor", 22)):
        case = synth.make_case("vlp16", n_keyframes=5, seed=seed, kind=kind, device="cpu")
        q = case["queries"][0]
        scan = q["scan"][::2].copy()
        out = {}
        for jm in (0, 1):
            cfg = orc.default_config(knn_mode=0, jacobian_mode=jm)        # brute-force k-NN: the ground truth
            pose, res, matP, corr = orc.scan2map(cfg, scan, case["map"], q["pose_init"], corr_iter=0)
            sfx = "" if jm == 0 else "_exactjac"
            out.update({f"pose{sfx}": pose, f"iters{sfx}": res.iters, f"converged{sfx}": res.converged,
                        f"is_degenerate{sfx}": res.is_degenerate, f"n_corr_iter{sfx}": np.array(res.n_corr_iter),
                        f"matP{sfx}": matP, f"AtA{sfx}": np.array(res.AtA, np.float32),
                        f"pose_iter{sfx}": np.array(res.pose_iter, np.float32)})
            if jm == 0:
                out.update(flag0=corr[0], coeff0=corr[1], nn0=corr[2])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), scan=scan, map=case["map"],
                            pose_init=q["pose_init"], pose_true=q["pose_true"], **out)
        print(name, scan.shape, case["map"].shape, out["pose"], out["iters"], out["is_degenerate"])
    # ---- EXTENSION: corner + surf registration, upstream LIO-SAM semantics
    case = synth.add_corners(synth.make_case("vlp16", n_keyframes=5, seed=23, device="cpu"), "vlp16", seed=23)
    q = case["queries"][0]
    scan = q["scan"][::2].copy()
    cfg = orc.default_config(knn_mode=0)
    pose, res, matP, corr = orc.scan2map_cs(cfg, q["corners"], case["corner_map"], scan, case["map"], q["pose_init"], corr_iter=0)
    np.savez_compressed(os.path.join(HERE, "s2m_corner.npz"), scan=scan, map=case["map"], corners=q["corners"],
                        corner_map=case["corner_map"], pose_init=q["pose_init"], pose_true=q["pose_true"],
                        pose=pose, iters=res.iters, n_corr_iter=np.array(res.n_corr_iter), matP=matP,
                        AtA=np.array(res.AtA, np.float32), cflag0=corr[0], ccoeff0=corr[1], cnn0=corr[2])
    print("s2m_corner", scan.shape, q["corners"].shape, case["corner_map"].shape, pose, res.iters)
    # ---- deskew + curvature
    boxes = synth.make_scene(5, length=60.0)
    sc = synth.cast_scan(boxes, [0.01, -0.02, 0.3, 10.0, 0.2, synth.SENSOR_HEIGHT], "vlp16", seed=3,
                         omega=(0.3, -0.2, 1.1), device="cpu")
    sel = slice(0, None, 5)
    t0 = 1700000000.25
    stamp = t0 - 0.013 + np.arange(90) * 0.002
    gyro = np.array([0.3, -0.2, 1.1]) + np.random.default_rng(1).normal(0, 0.05, (90, 3))
    imu = orc.imu_deskew_info(stamp, gyro, t0, t0 + 0.1)
    d = DeskewConfig(N_SCAN=16, downsampleRate=2, point_filter_num=3, lidarMinFront=1.0, lidarMinBack=5.0,
                     lidarMinLeft=2.0, lidarMinRight=2.0, lidarMaxRange=60.0, lidarMaxIntensity=90.0,
                     deskew_flag=1, imu_available=1, trig_mode=0)
    xyz, inten, ring, time = sc["xyz"][sel], sc["intensity"][sel], sc["ring"][sel], sc["time"][sel]
    out, keep = orc.project_point_cloud(d, xyz[:, 0], xyz[:, 1], xyz[:, 2], inten, ring, time, t0, imu)
    rng_ = sc["range"][:4000]
    curv, picked, label = orc.calculate_smoothness(rng_)
    np.savez_compressed(os.path.join(HERE, "prepare.npz"), xyz=xyz, intensity=inten, ring=ring, time=time,
                        t0=t0, stamp=stamp, gyro=gyro, imu_cur=imu[0], imu_T=imu[1][:imu[0] + 1],
                        imu_RX=imu[2][:imu[0] + 1], imu_RY=imu[3][:imu[0] + 1], imu_RZ=imu[4][:imu[0] + 1],
                        deskew_out=out, deskew_keep=keep, range=rng_, curvature=curv)
    print("prepare", xyz.shape, out.shape, curv[5:8])


if __name__ == "__main__":
    main()
