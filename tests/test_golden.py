"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py
from the synthetic generator + the brute-force CPU oracle; the reference itself
ships no fixtures).  CPU: the oracle (kd-tree, any thread count) reproduces them.
GPU: the HIP path reproduces them through the C ABI."""
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["s2m_street", "s2m_corridor"]


def _load(name):
    return np.load(os.path.join(HERE, name + ".npz"))


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(oracle, name):
    g = _load(name)
    for jm, sfx in ((0, ""), (1, "_exactjac")):
        cfg = oracle.default_config(knn_mode=1, n_threads=4, jacobian_mode=jm)
        pose, res, matP, corr = oracle.scan2map(cfg, g["scan"], g["map"], g["pose_init"], corr_iter=0)
        np.testing.assert_array_equal(pose, g["pose" + sfx])
        assert res.iters == int(g["iters" + sfx]) and res.is_degenerate == int(g["is_degenerate" + sfx])
        np.testing.assert_array_equal(np.array(res.n_corr_iter), g["n_corr_iter" + sfx])
        np.testing.assert_array_equal(matP, g["matP" + sfx])
        if jm == 0:
            np.testing.assert_array_equal(corr[0], g["flag0"])
            np.testing.assert_array_equal(corr[1], g["coeff0"])
            np.testing.assert_array_equal(corr[2], g["nn0"])
    assert int(_load("s2m_corridor")["is_degenerate"]) == 1 and int(_load("s2m_street")["is_degenerate"]) == 0


def test_oracle_reproduces_corner_golden(oracle):
    """Extension: brute-force fixture vs kd-tree / threaded oracle."""
    g = _load("s2m_corner")
    cfg = oracle.default_config(knn_mode=1, n_threads=4)
    pose, res, matP, corr = oracle.scan2map_cs(cfg, g["corners"], g["corner_map"], g["scan"], g["map"], g["pose_init"], corr_iter=0)
    np.testing.assert_array_equal(pose, g["pose"])
    assert res.iters == int(g["iters"])
    np.testing.assert_array_equal(np.array(res.n_corr_iter), g["n_corr_iter"])
    for a, k in zip(corr, ("cflag0", "ccoeff0", "cnn0")):
        np.testing.assert_array_equal(a, g[k])
    assert np.abs(pose - g["pose_true"]).max() < 0.05


def test_oracle_prepare_golden(oracle):
    import oracle.oracle as om
    g = _load("prepare")
    imu = oracle.imu_deskew_info(g["stamp"], g["gyro"], float(g["t0"]), float(g["t0"]) + 0.1)
    assert imu[0] == int(g["imu_cur"])
    np.testing.assert_array_equal(imu[1][:imu[0] + 1], g["imu_T"])
    np.testing.assert_array_equal(imu[4][:imu[0] + 1], g["imu_RZ"])
    d = om.DeskewConfig(N_SCAN=16, downsampleRate=2, point_filter_num=3, lidarMinFront=1.0, lidarMinBack=5.0,
                        lidarMinLeft=2.0, lidarMinRight=2.0, lidarMaxRange=60.0, lidarMaxIntensity=90.0,
                        deskew_flag=1, imu_available=1, trig_mode=0)
    out, keep = oracle.project_point_cloud(d, g["xyz"][:, 0], g["xyz"][:, 1], g["xyz"][:, 2], g["intensity"],
                                           g["ring"], g["time"], float(g["t0"]), imu)
    np.testing.assert_array_equal(out, g["deskew_out"])
    np.testing.assert_array_equal(keep, g["deskew_keep"])
    np.testing.assert_array_equal(oracle.calculate_smoothness(g["range"])[0], g["curvature"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_reproduces_golden(pkg, name):
    g = _load(name)
    matp_exact = []
    for jm, sfx in ((0, ""), (1, "_exactjac")):
        s2m = pkg.ScanToMap(record_corr_iter=0, jacobian_mode=jm)
        s2m.set_map(g["map"])
        pose, res, rc = s2m.scan2MapOptimization(g["scan"], g["pose_init"])
        assert res.iters == int(g["iters" + sfx]) and res.is_degenerate == int(g["is_degenerate" + sfx])
        assert list(res.n_corr_iter) == list(g["n_corr_iter" + sfx])
        np.testing.assert_allclose(pose[3:], g["pose" + sfx][3:], atol=1e-5)      # metres
        np.testing.assert_allclose(pose[:3], g["pose" + sfx][:3], atol=1e-6)      # radians
        # matP = V^-1 * V2 (MO:1807) from the same normal matrix: the eigen / inverse / product chain is bit-exact
        np.testing.assert_allclose(np.array(res.matP, np.float32).reshape(6, 6), g["matP" + sfx], atol=2e-5)
        matp_exact.append(bool(np.array_equal(np.array(res.matP, np.float32).reshape(6, 6), g["matP" + sfx])))
        if jm == 0:
            flag, coeff, nn = s2m.get_correspondences(0)
            np.testing.assert_array_equal(flag, g["flag0"])                       # bit-exact sets
            np.testing.assert_array_equal(nn, g["nn0"])
            np.testing.assert_array_equal(coeff[flag == 1].view(np.uint32), g["coeff0"][flag == 1].view(np.uint32))
        s2m.close()
    assert all(matp_exact), "matP differs from the oracle in the last bits"


@pytest.mark.gpu
def test_gpu_reproduces_corner_golden(pkg):
    g = _load("s2m_corner")
    s2m = pkg.ScanToMap(record_corr_iter=0)
    s2m.set_map(g["map"])
    s2m.set_corner_map(g["corner_map"])
    pose, res, rc = s2m.scan2MapOptimizationCS(g["corners"], g["scan"], g["pose_init"])
    assert rc == 0 and res.iters == int(g["iters"])
    assert list(res.n_corr_iter) == list(g["n_corr_iter"])
    np.testing.assert_allclose(pose[3:], g["pose"][3:], atol=1e-5)
    np.testing.assert_allclose(pose[:3], g["pose"][:3], atol=1e-6)
    flag, coeff, nn = s2m.get_corner_correspondences(0)
    np.testing.assert_array_equal(flag, g["cflag0"])
    np.testing.assert_array_equal(nn, g["cnn0"])
    gated = nn[:, 0] >= 0
    np.testing.assert_array_equal(coeff[gated].view(np.uint32), g["ccoeff0"][gated].view(np.uint32))
    s2m.close()


@pytest.mark.gpu
def test_gpu_prepare_golden(pkg):
    g = _load("prepare")
    imu = pkg.imu_deskew_info(g["stamp"], g["gyro"], float(g["t0"]), float(g["t0"]) + 0.1)
    d = pkg.deskew_default_config(N_SCAN=16, downsampleRate=2, point_filter_num=3, lidarMaxRange=60.0,
                                  lidarMaxIntensity=90.0)
    rec = pkg.pack_xyzirt(g["xyz"], g["intensity"], g["ring"], g["time"])
    out = pkg.deskew(d, rec, float(g["t0"]), imu)
    np.testing.assert_array_equal(out.view(np.uint32), g["deskew_out"].view(np.uint32))
    np.testing.assert_array_equal(pkg.curvature(g["range"])[0].view(np.uint32), g["curvature"].view(np.uint32))
