"""The C++ mirror of the reference's member functions (include/liogpu.hpp) driven by a
stand-in for the patched node (examples/s2m_node_stub.cpp, plain g++): same C ABI, same
result as the Python harness, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_node_stub_matches_python_path(pkg, tmp_path):
    g = np.load(os.path.join(ROOT, "tests", "golden", "s2m_street.npz"))
    exe = str(tmp_path / "s2m_node_stub")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "s2m_node_stub.cpp"),
                           "-L", os.path.join(ROOT, "lio-slam_amd"), "-lliogpu",
                           "-Wl,-rpath," + os.path.join(ROOT, "lio-slam_amd"), "-o", exe])
    scan_f, map_f = str(tmp_path / "scan.f32"), str(tmp_path / "map.f32")
    g["scan"].astype(np.float32).tofile(scan_f)
    g["map"].astype(np.float32).tofile(map_f)
    out = subprocess.check_output([exe, scan_f, map_f] + [repr(float(v)) for v in g["pose_init"]], text=True)
    tok = out.split()
    assert int(tok[1]) == int(g["iters"]) and int(tok[3]) == 1 and int(tok[5]) == 0
    pose_cpp = np.array([float(v) for v in tok[tok.index("pose") + 1:tok.index("pose") + 7]], np.float32)
    s2m = pkg.ScanToMap()
    s2m.set_map(g["map"])
    pose_py, res, _ = s2m.scan2MapOptimization(g["scan"], g["pose_init"])
    pose_py = pkg.transform_update(pose_py)           # the stub also runs transformUpdate (MO:1861)
    np.testing.assert_array_equal(pose_cpp, pose_py)
    np.testing.assert_allclose(pose_cpp[3:], g["pose"][3:], atol=1e-5)
    s2m.close()
    # the PointCloud2 entry point and the multi-device handle behind the same C++ members
    lines = out.strip().split("\n")
    pc2 = lines[1].split()
    assert pc2[0] == "pc2" and int(pc2[2]) == int(g["iters"])
    np.testing.assert_array_equal(np.array([float(v) for v in pc2[4:10]], np.float32), pose_py)
    rw = lines[2].split()                             # the device chain (downsample + register) behind the C++ member
    assert rw[0] == "raw" and int(rw[8]) == 1 and 0.7 * int(rw[6]) <= int(rw[4]) <= int(rw[6])
    mu = lines[3].split()
    assert mu[0] == "multi" and int(mu[2]) == int(g["iters"])
    pm = np.array([float(v) for v in mu[4:10]], np.float32)
    assert np.abs(pm[3:] - pose_py[3:]).max() <= 1e-5 and np.abs(pm[:3] - pose_py[:3]).max() <= 1e-6
