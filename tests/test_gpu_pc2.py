"""cloud_info wire path: sensor_msgs/PointCloud2 `data` blobs read in place, and keyframes that
never leave the device.  Reference: pcl::fromROSMsg(msgIn->cloud_deskewed, ...) MO:440, cloud_info.msg:27,
publishCloud / pcl::toROSMsg UT:369-379, cachePointCloud's per-sensor conversions IP:226-285, and
saveKeyFramesAndFactor's surfCloudKeyFrames.push_back MO:2136-2142."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _blob(xyz, intensity, point_step, off_x, off_i):
    """PointCloud2 data blob with x,y,z at off_x and intensity at off_i (other bytes = junk that must be ignored)."""
    n = len(xyz)
    b = np.full((n, point_step), 0xA5, np.uint8)
    b[:, off_x:off_x + 12] = np.ascontiguousarray(xyz, np.float32).view(np.uint8).reshape(n, 12)
    if off_i >= 0:
        b[:, off_i:off_i + 4] = np.ascontiguousarray(intensity, np.float32).view(np.uint8).reshape(n, 4)
    return b


@pytest.mark.parametrize("point_step,off_x,off_i,pin", [(32, 0, 16, 0), (32, 0, 16, 1), (16, 0, 12, 0), (20, 4, 16, 1), (48, 8, 24, 0)])
def test_register_on_a_pointcloud2_blob(pkg, small_case, point_step, off_x, off_i, pin):
    q = small_case["queries"][1]
    ref = pkg.ScanToMap(record_corr_iter=0)
    ref.set_map(small_case["map"])
    p0, r0, rc0 = ref.scan2MapOptimization(q["scan"], q["pose_init"])
    c0 = ref.get_correspondences(0)
    blob = _blob(q["scan"], np.arange(len(q["scan"]), dtype=np.float32), point_step, off_x, off_i)
    lay = pkg.PC2Layout(point_step=point_step, off_x=off_x, off_intensity=off_i, off_ring=-1, off_time=-1, pin_host=pin)
    s = pkg.ScanToMap(record_corr_iter=0)
    s.set_map(small_case["map"])
    p1, r1, rc1 = s.scan2MapOptimizationPC2(blob, len(q["scan"]), lay, q["pose_init"])
    c1 = s.get_correspondences(0)
    assert rc0 == rc1 == 0 and r0.iters == r1.iters
    np.testing.assert_array_equal(p0, p1)
    np.testing.assert_array_equal(np.array(r0.AtA, np.float32).view(np.uint32), np.array(r1.AtA, np.float32).view(np.uint32))
    for a, b in zip(c0, c1):
        np.testing.assert_array_equal(a, b)
    # malformed layouts are refused, not guessed
    bad = pkg.PC2Layout(point_step=point_step, off_x=point_step - 8, off_intensity=-1, off_ring=-1, off_time=-1)
    with pytest.raises(pkg.LioError):
        s.scan2MapOptimizationPC2(blob, len(q["scan"]), bad, q["pose_init"])
    ref.close(); s.close()


def _sweep(synth):
    boxes = synth.make_scene(41, length=60.0)
    return synth.cast_scan(boxes, synth.keyframe_poses(2, seed=41)[1], "vlp16", seed=301, omega=(0.02, -0.01, 0.15), device="cpu")


def _imu(lib, t0):
    stamp = t0 - 0.011 + np.arange(70) * 0.002
    gyro = np.tile(np.array([(0.02, -0.01, 0.15)]), (70, 1))
    return lib.imu_deskew_info(stamp, gyro, t0, t0 + 0.1)


def _oracle_deskew(oracle, sc, time_f32, t0, n_scan=16, deskew=1):
    import oracle.oracle as om
    d = om.DeskewConfig(N_SCAN=n_scan, downsampleRate=1, point_filter_num=1, lidarMinFront=0.0, lidarMinBack=0.0, lidarMinLeft=0.0,
                        lidarMinRight=0.0, lidarMaxRange=1000.0, lidarMaxIntensity=1.0e9, deskew_flag=deskew, imu_available=1, trig_mode=0)
    out, _ = oracle.project_point_cloud(d, sc["xyz"][:, 0], sc["xyz"][:, 1], sc["xyz"][:, 2], sc["intensity"],
                                        sc["ring"], np.ascontiguousarray(time_f32, np.float32), t0, _imu(oracle, t0))
    return out


def test_deskew_on_pointcloud2_blobs_of_the_four_sensor_layouts(pkg, oracle, synth):
    """IP:226-285 converts Ouster / Mulran / Robosense clouds to the Velodyne layout on the host before
    projectPointCloud; lio_deskew_pc2 reads each layout in place with the same conversions."""
    sc = _sweep(synth)
    n = len(sc["xyz"])
    t0 = 100.0
    dg = pkg.deskew_default_config(N_SCAN=16, point_filter_num=1, lidarMinFront=0, lidarMinBack=0, lidarMinLeft=0, lidarMinRight=0,
                                   lidarMaxRange=1000.0, lidarMaxIntensity=1.0e9)
    # --- Velodyne (IP:4-15): identical to the record entry point
    rec = pkg.pack_xyzirt(sc["xyz"], sc["intensity"], sc["ring"], sc["time"])
    want = pkg.deskew(dg, rec, t0, _imu(pkg, t0))
    lay = pkg.PC2Layout(point_step=32, off_x=0, off_intensity=16, off_ring=20, ring_type=4, off_time=24, time_type=0, pin_host=1)
    got = pkg.deskew_pc2(dg, rec.view(np.uint8), n, lay, t0, _imu(pkg, t0))
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    np.testing.assert_array_equal(want.view(np.uint32), _oracle_deskew(oracle, sc, sc["time"], t0).view(np.uint32))
    # --- Ouster (IP:17-31): x y z pad | intensity | uint32 t (ns) | uint16 reflectivity | uint8 ring | ... , point_step 48
    t_ns = np.round(sc["time"].astype(np.float64) * 1e9).astype(np.uint32)
    ous = np.dtype({"names": ["x", "y", "z", "intensity", "t", "reflectivity", "ring", "noise", "range"],
                    "formats": ["<f4", "<f4", "<f4", "<f4", "<u4", "<u2", "u1", "<u2", "<u4"],
                    "offsets": [0, 4, 8, 16, 20, 24, 26, 28, 32], "itemsize": 48})
    o = np.zeros(n, ous)
    o["x"], o["y"], o["z"], o["intensity"], o["t"], o["ring"] = sc["xyz"][:, 0], sc["xyz"][:, 1], sc["xyz"][:, 2], sc["intensity"], t_ns, sc["ring"]
    lay = pkg.PC2Layout(point_step=48, off_x=0, off_intensity=16, off_ring=26, ring_type=2, off_time=20, time_type=1)
    got = pkg.deskew_pc2(dg, o.view(np.uint8), n, lay, t0, _imu(pkg, t0))
    time_conv = t_ns.astype(np.float32) * np.float32(1e-9)                                   # dst.time = src.t * 1e-9f, IP:243
    np.testing.assert_array_equal(got.view(np.uint32), _oracle_deskew(oracle, sc, time_conv, t0).view(np.uint32))
    # --- Mulran (IP:47-58): uint32 t taken as is, int ring
    t_raw = (sc["time"] * 1000).astype(np.uint32)                                            # ticks; (float)t is what the reference uses
    mul = np.dtype({"names": ["x", "y", "z", "intensity", "t", "ring"], "formats": ["<f4", "<f4", "<f4", "<f4", "<u4", "<i4"],
                    "offsets": [0, 4, 8, 16, 20, 24], "itemsize": 32})
    m = np.zeros(n, mul)
    m["x"], m["y"], m["z"], m["intensity"], m["t"], m["ring"] = sc["xyz"][:, 0], sc["xyz"][:, 1], sc["xyz"][:, 2], sc["intensity"], t_raw, sc["ring"]
    lay = pkg.PC2Layout(point_step=32, off_x=0, off_intensity=16, off_ring=24, ring_type=5, off_time=20, time_type=2)
    got = pkg.deskew_pc2(dg, m.view(np.uint8), n, lay, t0, _imu(pkg, t0))
    np.testing.assert_array_equal(got.view(np.uint32), _oracle_deskew(oracle, sc, t_raw.astype(np.float32), t0).view(np.uint32))
    # --- Robosense (IP:33-45): double absolute stamps, time = stamp - stamp of point 0
    stamp = 1.7e9 + sc["time"].astype(np.float64)
    rob = np.dtype({"names": ["x", "y", "z", "intensity", "ring", "timestamp"], "formats": ["<f4", "<f4", "<f4", "<f4", "<u2", "<f8"],
                    "offsets": [0, 4, 8, 16, 20, 24], "itemsize": 32})
    r = np.zeros(n, rob)
    r["x"], r["y"], r["z"], r["intensity"], r["ring"], r["timestamp"] = sc["xyz"][:, 0], sc["xyz"][:, 1], sc["xyz"][:, 2], sc["intensity"], sc["ring"], stamp
    lay = pkg.PC2Layout(point_step=32, off_x=0, off_intensity=16, off_ring=20, ring_type=4, off_time=24, time_type=3)
    got = pkg.deskew_pc2(dg, r.view(np.uint8), n, lay, t0, _imu(pkg, t0))
    np.testing.assert_array_equal(got.view(np.uint32), _oracle_deskew(oracle, sc, (stamp - stamp[0]).astype(np.float32), t0).view(np.uint32))
    # --- no per-point time field: deskewFlag = -1 (IP:341-356), points pass through un-rotated
    lay = pkg.PC2Layout(point_step=32, off_x=0, off_intensity=16, off_ring=20, ring_type=4, off_time=-1, time_type=0)
    got = pkg.deskew_pc2(dg, rec.view(np.uint8), n, lay, t0, _imu(pkg, t0))
    np.testing.assert_array_equal(got.view(np.uint32), _oracle_deskew(oracle, sc, sc["time"], t0, deskew=-1).view(np.uint32))


def test_keyframes_never_leave_the_device(pkg, synth):
    """A short drive where every registered scan becomes a keyframe straight from the handle's staged records
    (MO:2136-2142) and the next local map is assembled from the resident keyframes: same maps, same poses as the chain
    that adds every keyframe from the host."""
    boxes = synth.make_scene(43, length=60.0)
    poses = synth.keyframe_poses(6, spacing=0.8, seed=43)
    clouds = []
    for k, p in enumerate(poses):
        sc = synth.cast_scan(boxes, p, "vlp16", seed=500 + k, device="cpu")
        ds, _ = pkg.voxel_grid(np.concatenate([sc["xyz"], sc["intensity"][:, None]], 1), 0.4)
        clouds.append(ds)                                       # [n,4] xyzi, laserCloudSurfLastDS

    def pcl(c):
        rec = np.zeros((len(c), 8), np.float32)
        rec[:, :3], rec[:, 3], rec[:, 4] = c[:, :3], 1.0, c[:, 3]
        return rec

    lay = pkg.PC2Layout(point_step=32, off_x=0, off_intensity=16, off_ring=-1, off_time=-1)
    trajs, maps = [], []
    for on_device in (False, True):
        s2m, store = pkg.ScanToMap(), pkg.KeyframeStore()
        ids, kposes, traj, sizes = [], [], [], []
        for k, c in enumerate(clouds):
            if k == 0:
                pose = poses[0].astype(np.float32)
                kid = store.add(c)                              # the first keyframe has no registration behind it
            else:
                _, n_map, _ = store.assemble(ids, np.stack(kposes), 0.5, s2m=s2m, want_output=False)
                sizes.append(n_map)
                guess = (traj[-1] + (traj[-1] - traj[-2] if k > 1 else 0)).astype(np.float32)
                pose, res, rc = s2m.scan2MapOptimizationPC2(pcl(c).view(np.uint8), len(c), lay, guess)
                assert rc == 0
                kid = store.add_from_handle(s2m, 0) if on_device else store.add(c)
            ids.append(kid); kposes.append(np.asarray(pose, np.float32)); traj.append(np.asarray(pose, np.float32))
        out, n_out, _ = store.assemble(ids, np.stack(kposes), 0.5, want_output=True, max_out=200000)
        trajs.append(np.stack(traj)); maps.append(out)
        assert len(store) == len(clouds)
        store.close(); s2m.close()
    np.testing.assert_array_equal(trajs[0], trajs[1])
    np.testing.assert_array_equal(maps[0].view(np.uint32), maps[1].view(np.uint32))     # xyz AND intensity of the final map
    assert np.abs(trajs[1][:, 3:] - poses[:, 3:].astype(np.float32)).max() < 0.05
