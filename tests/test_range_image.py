"""EXTENSION beyond this reference (north_star's "range-image build"): projectPointCloud +
cloudExtraction of upstream LIO-SAM on this fork's deskewPoint.  No reference fixture exists (parity
unpinned): known-answer checks of the CPU restatement, and the HIP path bit-exact against it."""
import importlib

import numpy as np
import pytest

synth = importlib.import_module("lio-slam_amd.synth")
OMEGA = (0.1, -0.05, 0.5)


def _sweep(sensor="vlp16", seed=3, omega=OMEGA):
    boxes = synth.make_scene(5, length=60.0)
    return synth.cast_scan(boxes, [0.0, 0.0, 0.3, 10.0, 0.2, synth.SENSOR_HEIGHT], sensor, seed=seed, device="cpu", omega=omega)


def _imu(lib, t0, omega=OMEGA):
    stamp = t0 - 0.011 + np.arange(70) * 0.002
    return lib.imu_deskew_info(stamp, np.tile(np.array([omega]), (70, 1)), t0, t0 + 0.1)


def _ocfg(n_scan, **kw):
    import oracle.oracle as om
    d = dict(N_SCAN=n_scan, downsampleRate=1, point_filter_num=1, lidarMinFront=0.0, lidarMinBack=0.0, lidarMinLeft=0.0,
             lidarMinRight=0.0, lidarMaxRange=1000.0, lidarMaxIntensity=1e9, deskew_flag=1, imu_available=1, trig_mode=0)
    d.update(kw)
    return om.DeskewConfig(**d)


def test_oracle_range_image_structure(oracle):
    sc = _sweep()
    t0 = 50.0
    r = oracle.range_image(_ocfg(16), 1800, 1.0, sc["xyz"], sc["intensity"], sc["ring"], sc["time"], t0, _imu(oracle, t0))
    n = len(r["cloud"])
    assert n == len(sc["xyz"])                                   # the generator fires once per (ring, column) cell
    assert r["start_ring"][0] == 4                               # 0 - 1 + 5
    first = r["start_ring"] - 4
    last = r["end_ring"] + 5
    assert first[0] == 0 and last[-1] == n - 1 and np.all(first[1:] == last[:-1] + 1)
    for i in range(16):
        assert np.all(np.diff(r["col"][first[i]:last[i] + 1]) > 0)           # ascending column inside a ring
    # pointRange is the RAW range; the cloud is deskewed: the same points as this fork's projectPointCloud produces
    d = _ocfg(16)
    ref, keep = oracle.project_point_cloud(d, sc["xyz"][:, 0], sc["xyz"][:, 1], sc["xyz"][:, 2], sc["intensity"], sc["ring"],
                                           sc["time"], t0, _imu(oracle, t0))
    # the generator fires at azimuth 2*pi*col/H from +x; upstream's rule measures from +y and adds H/2:
    # image column = generator column + H/2 (mod H), a fixed rotation of the image
    pred = (sc["col"] + 900) % 1800
    order = np.lexsort((pred, sc["ring"]))
    np.testing.assert_array_equal(r["col"], pred[order])
    np.testing.assert_allclose(r["range"], np.linalg.norm(sc["xyz"][order], axis=1), rtol=1e-6)
    np.testing.assert_array_equal(r["cloud"], ref[order])        # same deskew as this fork's projectPointCloud, ring-major order


def test_oracle_range_image_first_point_wins_and_filters(oracle):
    sc = _sweep(omega=(0, 0, 0))
    none = (0, np.zeros(2000), np.zeros(2000), np.zeros(2000), np.zeros(2000))
    xyz = np.concatenate([sc["xyz"], sc["xyz"][:500] * 1.01])   # 500 later points landing in occupied cells
    inten = np.concatenate([sc["intensity"], np.full(500, 7.0, np.float32)])
    ring = np.concatenate([sc["ring"], sc["ring"][:500]])
    time = np.concatenate([sc["time"], sc["time"][:500]])
    d = _ocfg(16, imu_available=0)
    a = oracle.range_image(d, 1800, 1.0, sc["xyz"], sc["intensity"], sc["ring"], sc["time"], 0.0, none)
    b = oracle.range_image(d, 1800, 1.0, xyz, inten, ring, time, 0.0, none)
    for k in ("cloud", "col", "range", "start_ring", "end_ring"):
        np.testing.assert_array_equal(a[k], b[k])
    # range gate and ring stride
    c = oracle.range_image(_ocfg(16, imu_available=0, downsampleRate=2, lidarMaxRange=20.0), 1800, 5.0,
                           sc["xyz"], sc["intensity"], sc["ring"], sc["time"], 0.0, none)
    assert np.all((c["range"] >= 5.0) & (c["range"] <= 20.0))
    assert np.all(c["end_ring"][1::2] - c["start_ring"][1::2] == -10)          # odd rings are empty
    empty = oracle.range_image(d, 1800, 1.0, sc["xyz"][:0], sc["intensity"][:0], sc["ring"][:0], sc["time"][:0], 0.0, none)
    assert len(empty["cloud"]) == 0 and np.all(empty["start_ring"] == 4) and np.all(empty["end_ring"] == -6)


@pytest.mark.gpu
@pytest.mark.parametrize("sensor,n_scan,H", [("vlp16", 16, 1800), ("hdl64", 64, 1800), ("os1_128", 128, 2048)])
def test_gpu_range_image_bit_exact(pkg, oracle, sensor, n_scan, H):
    sc = _sweep(sensor)
    t0 = 50.0
    rec = pkg.pack_xyzirt(sc["xyz"], sc["intensity"], sc["ring"], sc["time"])
    g = pkg.range_image(rec, t0, _imu(pkg, t0), N_SCAN=n_scan, Horizon_SCAN=H)
    o = oracle.range_image(_ocfg(n_scan), H, 1.0, sc["xyz"], sc["intensity"], sc["ring"], sc["time"], t0, _imu(oracle, t0))
    assert len(g["cloud"]) == len(o["cloud"]) > 1000
    for k in ("start_ring", "end_ring", "col"):
        np.testing.assert_array_equal(g[k], o[k])
    assert np.array_equal(g["range"].view(np.uint32), o["range"].view(np.uint32))
    assert np.array_equal(g["cloud"].view(np.uint32), o["cloud"].view(np.uint32))
    # and it feeds the feature extraction: the whole front end on the GPU equals the whole front end on the CPU
    fg = pkg.extract_features(g["cloud"], g["start_ring"], g["end_ring"], g["col"], g["range"])
    fo = oracle.extract_features(o["cloud"], o["start_ring"], o["end_ring"], o["col"], o["range"])
    assert np.array_equal(fg["corner"].view(np.uint32), fo["corner"].view(np.uint32))
    assert np.array_equal(fg["surface"].view(np.uint32), fo["surface"].view(np.uint32))


@pytest.mark.gpu
def test_gpu_range_image_duplicates_filters_and_empty(pkg, oracle):
    sc = _sweep(omega=(0, 0, 0))
    none = (0, np.zeros(2000), np.zeros(2000), np.zeros(2000), np.zeros(2000))
    xyz = np.concatenate([sc["xyz"][:500] * 1.01, sc["xyz"]])   # the EARLIER duplicates win their cells
    inten = np.concatenate([np.full(500, 7.0, np.float32), sc["intensity"]])
    ring = np.concatenate([sc["ring"][:500], sc["ring"]])
    time = np.concatenate([sc["time"][:500], sc["time"]])
    rec = pkg.pack_xyzirt(xyz, inten, ring, time)
    g = pkg.range_image(rec, 0.0, none, N_SCAN=16, Horizon_SCAN=1800, downsampleRate=2, lidarMinRange=3.0, lidarMaxRange=40.0)
    o = oracle.range_image(_ocfg(16, imu_available=0, downsampleRate=2, lidarMaxRange=40.0), 1800, 3.0, xyz, inten, ring, time, 0.0, none)
    for k in ("start_ring", "end_ring", "col"):
        np.testing.assert_array_equal(g[k], o[k])
    assert np.array_equal(g["cloud"].view(np.uint32), o["cloud"].view(np.uint32))
    assert (g["cloud"][:, 3] == 7.0).sum() > 100
    e = pkg.range_image(rec[:0], 0.0, none, N_SCAN=16, Horizon_SCAN=1800)
    assert len(e["cloud"]) == 0 and np.all(e["start_ring"] == 4) and np.all(e["end_ring"] == -6)
