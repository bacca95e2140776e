"""Oracle checks for the deskew path (IP:359-418, 502-615)."""
import ctypes as C

import numpy as np


def _dcfg(oracle_mod, **kw):
    c = oracle_mod.DeskewConfig(N_SCAN=16, downsampleRate=1, point_filter_num=1, lidarMinFront=0, lidarMinBack=0,
                                lidarMinLeft=0, lidarMinRight=0, lidarMaxRange=1000, lidarMaxIntensity=1000,
                                deskew_flag=1, imu_available=1, trig_mode=0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_imu_integration_and_window(oracle):
    stamp = 100.0 + np.arange(100) * 0.002           # 500 Hz
    gyro = np.tile(np.array([[0.1, -0.2, 0.3]]), (100, 1))
    cur, T, RX, RY, RZ = oracle.imu_deskew_info(stamp, gyro, 100.05, 100.15)
    first = np.nonzero(stamp >= 100.05 - 0.01)[0][0]
    last = np.nonzero(stamp <= 100.15 + 0.01)[0][-1]
    assert cur == last - first
    np.testing.assert_array_equal(T[:cur + 1], stamp[first:last + 1])
    np.testing.assert_allclose(RX[:cur + 1], 0.1 * (stamp[first:last + 1] - stamp[first]), atol=1e-12)
    np.testing.assert_allclose(RZ[:cur + 1], 0.3 * (stamp[first:last + 1] - stamp[first]), atol=1e-12)
    assert RX[0] == 0 and RY[0] == 0 and RZ[0] == 0
    cur, *_ = oracle.imu_deskew_info(stamp, gyro, 200.0, 200.1)      # queue entirely too old
    assert cur == 0
    cur, *_ = oracle.imu_deskew_info(stamp[:1], gyro[:1], 100.0, 100.1)   # a single sample: not available
    assert cur == 0


def test_find_rotation_interpolates(oracle):
    T = np.zeros(2000); RX = np.zeros(2000); RY = np.zeros(2000); RZ = np.zeros(2000)
    T[:4] = [0.0, 0.1, 0.2, 0.3]; RX[:4] = [0, 1, 2, 4]
    f = C.c_float; rx, ry, rz = f(), f(), f()
    def rot(t):
        oracle.lib.lo_find_rotation(t, T, RX, RY, RZ, 3, C.byref(rx), C.byref(ry), C.byref(rz))
        return rx.value
    assert rot(0.05) == np.float32(0.5)
    assert rot(0.25) == np.float32(3.0)
    assert rot(0.35) == np.float32(4.0)     # beyond the last sample: un-interpolated (IP:514)
    assert rot(-1.0) == np.float32(0.0)     # before the first: sample 0


def test_deskew_recovers_static_scan(oracle, synth):
    """A sensor spinning at a constant body rate: deskewing with the ideal gyro
    integral must reproduce the scan of a static sensor (same noise seed)."""
    import oracle.oracle as om
    boxes = synth.make_scene(5, length=60.0)
    pose = [0.0, 0.0, 0.3, 10.0, 0.2, synth.SENSOR_HEIGHT]
    omega = (0.2, -0.1, 0.8)
    moving = synth.cast_scan(boxes, pose, "vlp16", seed=9, omega=omega, device="cpu")
    static = synth.cast_scan(boxes, pose, "vlp16", seed=9, device="cpu")
    t0 = 50.0
    stamp = t0 - 0.004 + np.arange(80) * 0.002
    cur, T, RX, RY, RZ = oracle.imu_deskew_info(stamp, np.tile(np.array([omega]), (80, 1)), t0, t0 + 0.1)
    assert cur > 0
    # the table starts 4 ms before the sweep: shift so that angle(t0) = 0 like the generator
    out, keep = oracle.project_point_cloud(_dcfg(om), moving["xyz"][:, 0], moving["xyz"][:, 1], moving["xyz"][:, 2],
                                           moving["intensity"], moving["ring"], moving["time"], t0,
                                           (cur, T, RX, RY, RZ))
    assert len(out) == len(moving["xyz"])
    # expected: the point expressed in the sensor frame at the start of the sweep,
    # p_start = Rz(wz t) Ry(wy t) Rx(wx t) p_inst  (what the generator applied)
    t = moving["time"].astype(np.float64)
    exp = np.empty((len(t), 3))
    for i in range(len(t)):
        exp[i] = synth.rpy_matrix(omega[0] * t[i], omega[1] * t[i], omega[2] * t[i]) @ moving["xyz"][i].astype(np.float64)
    rng_ = np.linalg.norm(exp, axis=1)
    err = np.linalg.norm(out[:, :3] - exp, axis=1) / rng_
    # residual: the IMU table starts 4 ms before the sweep (rotations do not commute) + fp32
    assert err.max() < 5e-4, err.max()
    raw_err = np.linalg.norm(moving["xyz"] - exp, axis=1) / rng_
    assert raw_err.max() > 0.05                      # without deskew: |omega| * 0.1 s
    # and the deskewed cloud lands on the same surfaces as a static sensor's cloud
    Tw = synth.pose_matrix(pose)
    world = out[:, :3].astype(np.float64) @ Tw[:3, :3].T + Tw[:3, 3]
    ground = np.abs(world[:, 2]) < 0.08
    assert ground.mean() > 0.2                       # many returns lie on z = 0 after deskew
    assert len(static["xyz"]) > 0
    np.testing.assert_array_equal(out[:, 3], moving["intensity"])


def test_filters_and_order(oracle):
    import oracle.oracle as om
    n = 40
    x = np.linspace(-3, 3, n).astype(np.float32); y = np.zeros(n, np.float32) + 0.5; z = np.zeros(n, np.float32)
    inten = np.full(n, 10, np.float32); inten[7] = 200
    ring = (np.arange(n) % 20).astype(np.uint16)
    time = np.zeros(n, np.float32)
    imu = (0, np.zeros(2000), np.zeros(2000), np.zeros(2000), np.zeros(2000))
    cfg = _dcfg(om, lidarMinFront=1.0, lidarMinBack=5.0, lidarMinLeft=2.0, lidarMinRight=2.0,
                lidarMaxIntensity=100.0, downsampleRate=2, point_filter_num=3, imu_available=0)
    out, keep = oracle.project_point_cloud(cfg, x, y, z, inten, ring, time, 0.0, imu)
    exp = [i for i in range(n)
           if not ((y[i] < 1.0 and -5.0 < y[i] and x[i] < 2.0 and -2.0 < x[i]) or inten[i] > 100)
           and ring[i] < 16 and ring[i] % 2 == 0 and i % 3 == 0]
    assert list(keep) == exp and len(exp) > 0
    np.testing.assert_array_equal(out[:, 0], x[exp])      # no IMU: points pass through unchanged (IP:547)
