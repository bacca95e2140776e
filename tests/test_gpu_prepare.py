"""GPU parity of the two upstream per-point passes (through the C ABI):
K1 filter + IMU-rotation deskew (IP:545-615) and K2 range curvature (FE:81-101).
Bar: bit-exact against the CPU oracle, survivors in input order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_dcfg(om, g):
    return om.DeskewConfig(N_SCAN=g.N_SCAN, downsampleRate=g.downsampleRate, point_filter_num=g.point_filter_num,
                           lidarMinFront=g.lidarMinFront, lidarMinBack=g.lidarMinBack, lidarMinLeft=g.lidarMinLeft,
                           lidarMinRight=g.lidarMinRight, lidarMaxRange=g.lidarMaxRange,
                           lidarMaxIntensity=g.lidarMaxIntensity, deskew_flag=g.deskew_flag, imu_available=1, trig_mode=0)


def _scan(synth, omega, sensor="vlp16", seed=3):
    boxes = synth.make_scene(5, length=60.0)
    pose = [0.01, -0.02, 0.3, 10.0, 0.2, synth.SENSOR_HEIGHT]
    return synth.cast_scan(boxes, pose, sensor, seed=seed, omega=omega, device="cpu")


@pytest.mark.parametrize("cfg", [
    dict(),                                                              # reference defaults (UT:275-285)
    dict(point_filter_num=1, lidarMinFront=0, lidarMinBack=0, lidarMinLeft=0, lidarMinRight=0),
    dict(downsampleRate=2, point_filter_num=5, lidarMaxRange=30.0, lidarMaxIntensity=60.0),
    dict(N_SCAN=8, point_filter_num=2),
])
def test_deskew_bit_exact(pkg, oracle, synth, cfg):
    import oracle.oracle as om
    sc = _scan(synth, (0.3, -0.2, 1.1))
    t0 = 1700000000.25
    stamp = t0 - 0.013 + np.arange(90) * 0.002
    rng = np.random.default_rng(1)
    gyro = np.array([0.3, -0.2, 1.1]) + rng.normal(0, 0.05, (90, 3))
    imu_g = pkg.imu_deskew_info(stamp, gyro, t0, t0 + 0.1)
    imu_o = oracle.imu_deskew_info(stamp, gyro, t0, t0 + 0.1)
    assert imu_g[0] == imu_o[0] > 0
    for a, b in zip(imu_g[1:], imu_o[1:]):
        np.testing.assert_array_equal(a, b)
    g = pkg.deskew_default_config(**cfg)
    rec = pkg.pack_xyzirt(sc["xyz"], sc["intensity"], sc["ring"], sc["time"])
    out_g = pkg.deskew(g, rec, t0, imu_g)
    out_o, keep = oracle.project_point_cloud(_oracle_dcfg(om, g), sc["xyz"][:, 0], sc["xyz"][:, 1], sc["xyz"][:, 2],
                                             sc["intensity"], sc["ring"], sc["time"], t0, imu_o)
    assert len(out_g) == len(out_o) > 100
    np.testing.assert_array_equal(out_g.view(np.uint32), out_o.view(np.uint32))


def test_deskew_without_imu_or_time_passes_points_through(pkg, oracle, synth):
    import oracle.oracle as om
    sc = _scan(synth, (0.0, 0.0, 0.0))
    rec = pkg.pack_xyzirt(sc["xyz"], sc["intensity"], sc["ring"], sc["time"])
    none = (0, np.zeros(2000), np.zeros(2000), np.zeros(2000), np.zeros(2000))
    g = pkg.deskew_default_config(point_filter_num=1)
    out = pkg.deskew(g, rec, 0.0, none)                                   # imuAvailable == false, IP:547
    o = _oracle_dcfg(om, g); o.imu_available = 0
    out_o, keep = oracle.project_point_cloud(o, sc["xyz"][:, 0], sc["xyz"][:, 1], sc["xyz"][:, 2],
                                             sc["intensity"], sc["ring"], sc["time"], 0.0, none)
    np.testing.assert_array_equal(out, out_o)
    np.testing.assert_array_equal(out[:, :3], sc["xyz"][keep])
    g2 = pkg.deskew_default_config(point_filter_num=1, deskew_flag=-1)    # no per-point time field
    stamp = np.arange(60) * 0.002
    imu = pkg.imu_deskew_info(stamp, np.ones((60, 3)), 0.0, 0.1)
    np.testing.assert_array_equal(pkg.deskew(g2, rec, 0.0, imu), out)
    # everything filtered / empty input
    g3 = pkg.deskew_default_config(lidarMaxRange=0.0)
    assert len(pkg.deskew(g3, rec, 0.0, none)) == 0
    assert len(pkg.deskew(g, rec[:0], 0.0, none)) == 0


def test_deskew_large_cloud_order_preserved(pkg, oracle, synth):
    import oracle.oracle as om
    sc = _scan(synth, (0.1, 0.1, 0.5), sensor="hdl64", seed=5)           # ~100 k points, many workgroups
    t0 = 10.0
    stamp = t0 - 0.01 + np.arange(70) * 0.002
    gyro = np.tile(np.array([[0.1, 0.1, 0.5]]), (70, 1))
    imu = pkg.imu_deskew_info(stamp, gyro, t0, t0 + 0.1)
    g = pkg.deskew_default_config(N_SCAN=64, point_filter_num=3)
    rec = pkg.pack_xyzirt(sc["xyz"], sc["intensity"], sc["ring"], sc["time"])
    out_g = pkg.deskew(g, rec, t0, imu)
    out_o, keep = oracle.project_point_cloud(_oracle_dcfg(om, g), sc["xyz"][:, 0], sc["xyz"][:, 1], sc["xyz"][:, 2],
                                             sc["intensity"], sc["ring"], sc["time"], t0, imu)
    np.testing.assert_array_equal(out_g.view(np.uint32), out_o.view(np.uint32))
    np.testing.assert_array_equal(out_g[:, 3], sc["intensity"][keep])     # order == input order (IP:613)


@pytest.mark.parametrize("n", [0, 5, 10, 11, 12, 255, 256, 257, 266, 1000, 115200])
def test_curvature_bit_exact(pkg, oracle, n):
    rng = np.random.default_rng(n)
    r = rng.uniform(0.5, 100.0, n).astype(np.float32)
    cg, pg, lg = pkg.curvature(r)
    co, po, lo = oracle.calculate_smoothness(r)
    np.testing.assert_array_equal(cg.view(np.uint32), co.view(np.uint32))
    np.testing.assert_array_equal(pg, po)
    np.testing.assert_array_equal(lg, lo)
