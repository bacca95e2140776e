"""One mapping callback as ONE device chain (round-2 verdict, "what's missing" 1): downsampleCurrentScan MO:1605-1611
(`downSizeFilterSurf.filter(*laserCloudSurfLastDS)` on the whole deskewed cloud, MO:440/471) followed by
scan2MapOptimization MO:1839-1865, from the blob of cloud_info.cloud_deskewed, with no host round trip between the voxel
filter and the registration -- lio_s2m_register_raw.  It must be bit-identical to the two stand-alone calls
(lio_voxel_grid, then lio_s2m_register on its output), to the CPU oracle's voxel filter, and leave the filtered cloud
staged so that saveKeyFramesAndFactor MO:2136-2142 can append it as a keyframe without a copy through the host."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _raw_sweep(synth, sensor="vlp16", seed=301, k=1):
    boxes = synth.make_scene(11, length=60.0)
    pose = synth.keyframe_poses(6, seed=11)[k]
    sc = synth.cast_scan(boxes, pose, sensor, seed=seed, device="cpu")
    return sc, pose


def _pcl(xyz, intensity):
    rec = np.zeros((len(xyz), 8), np.float32)
    rec[:, :3], rec[:, 3], rec[:, 4] = xyz, 1.0, intensity
    return rec


@pytest.mark.parametrize("leaf", [0.4, 0.2])
@pytest.mark.parametrize("pin", [0, 1])
def test_downsample_and_register_is_one_device_chain(pkg, oracle, synth, small_case, leaf, pin):
    sc, pose = _raw_sweep(synth)
    guess = (pose + np.array([0.004, -0.003, 0.01, 0.08, -0.05, 0.03])).astype(np.float32)
    xyzi = np.concatenate([sc["xyz"], sc["intensity"][:, None]], 1).astype(np.float32)
    # the two stand-alone calls a host-side chain would make
    ds_ref, rc_v = pkg.voxel_grid(xyzi, leaf)
    assert rc_v == 0
    ref = pkg.ScanToMap(record_corr_iter=0)
    ref.set_map(small_case["map"])
    p0, r0, rc0 = ref.scan2MapOptimization(_pcl(ds_ref[:, :3], ds_ref[:, 3]), guess)
    c0 = ref.get_correspondences(0)
    # the oracle's pcl::VoxelGrid restatement agrees with the stand-alone filter (so the chain is pinned to it too)
    ds_o = oracle.voxel_grid(xyzi, leaf)[0] if hasattr(oracle, "voxel_grid") else None
    if ds_o is not None:
        np.testing.assert_array_equal(ds_ref.view(np.uint32), np.asarray(ds_o, np.float32).view(np.uint32))
    # the chain
    lay = pkg.PC2Layout(point_step=32, off_x=0, off_intensity=16, off_ring=-1, off_time=-1, pin_host=pin)
    h = pkg.ScanToMap(record_corr_iter=0)
    h.set_map(small_case["map"])
    p1, r1, rc1, ds1 = h.downsampleAndScan2MapOptimization(_pcl(sc["xyz"], sc["intensity"]), len(xyzi), lay, leaf, guess, want_ds=True)
    c1 = h.get_correspondences(0)
    assert rc0 == rc1 == 0 and r0.iters == r1.iters and r0.converged == r1.converged
    np.testing.assert_array_equal(ds_ref.view(np.uint32), ds1.view(np.uint32))              # laserCloudSurfLastDS, xyz and intensity
    np.testing.assert_array_equal(p0, p1)
    for f in ("AtA", "AtB", "matP"):
        np.testing.assert_array_equal(np.array(getattr(r0, f), np.float32).view(np.uint32), np.array(getattr(r1, f), np.float32).view(np.uint32))
    for a, b in zip(c0, c1):
        np.testing.assert_array_equal(a, b)
    # a second callback on the same handle re-uses the kept workspace (a smaller and a larger cloud)
    for cut in (3, 1):
        sub = xyzi[::cut]
        ds2, _ = pkg.voxel_grid(sub, leaf)
        pa, ra, _ = ref.scan2MapOptimization(_pcl(ds2[:, :3], ds2[:, 3]), guess)
        pb, rb, _, n_ds = h.downsampleAndScan2MapOptimization(_pcl(sub[:, :3], sub[:, 3]), len(sub), lay, leaf, guess)
        assert n_ds == len(ds2) and ra.iters == rb.iters
        np.testing.assert_array_equal(pa, pb)
    ref.close(); h.close()


def test_chain_from_a_cloud_that_already_lives_on_the_device(pkg, synth, small_case):
    """cloud_deskewed produced on the device (lio_deskew output kept in HBM): the blob is read in place, no H2D at all."""
    sc, pose = _raw_sweep(synth, seed=302, k=2)
    guess = (pose + np.array([0.0, 0.002, -0.01, 0.05, 0.06, -0.02])).astype(np.float32)
    rec = _pcl(sc["xyz"], sc["intensity"])
    lay = pkg.PC2Layout(point_step=32, off_x=0, off_intensity=16, off_ring=-1, off_time=-1)
    a = pkg.ScanToMap(); a.set_map(small_case["map"])
    b = pkg.ScanToMap(); b.set_map(small_case["map"])
    pa, ra, _, da = a.downsampleAndScan2MapOptimization(rec, len(rec), lay, 0.4, guess, want_ds=True)
    dev = pkg.DeviceBuffer(rec)                                       # (the library's own runtime: no second HIP runtime in the process)
    pb, rb, _, db = b.downsampleAndScan2MapOptimization(None, len(rec), lay, 0.4, guess, want_ds=True, device_ptr=dev.ptr)
    dev.close()
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(da.view(np.uint32), db.view(np.uint32))
    assert ra.iters == rb.iters
    a.close(); b.close()


def test_leaf_that_overflows_the_voxel_index_passes_the_cloud_through(pkg, synth, small_case):
    """mappingSurfLeafSize: 0.01 over a ~200 m extent (config/6t.yaml:112): PCL warns "Leaf size is too small" and copies the
    input; the reference then registers the UNFILTERED cloud, and so does the chain."""
    sc, pose = _raw_sweep(synth, seed=303, k=3)
    guess = pose.astype(np.float32)
    xyz = sc["xyz"].copy()
    xyz[0] = (9000.0, -9000.0, 5.0)                                   # one far return blows the bounding box up (1.8e6^2 x ... voxels of 1 cm)
    rec = _pcl(xyz, sc["intensity"])
    lay = pkg.PC2Layout(point_step=32, off_x=0, off_intensity=16, off_ring=-1, off_time=-1)
    ds, rc_v = pkg.voxel_grid(np.concatenate([xyz, sc["intensity"][:, None]], 1), 0.01)
    assert rc_v == 1 and len(ds) == len(xyz)
    h = pkg.ScanToMap(); h.set_map(small_case["map"])
    ref = pkg.ScanToMap(); ref.set_map(small_case["map"])
    p1, r1, rc1, n_ds = h.downsampleAndScan2MapOptimization(rec, len(rec), lay, 0.01, guess)
    p0, r0, rc0 = ref.scan2MapOptimization(rec, guess)
    assert n_ds == len(xyz) and rc0 == rc1 and r0.iters == r1.iters
    np.testing.assert_array_equal(p0, p1)
    h.close(); ref.close()


def test_keyframe_from_the_chain_keeps_its_intensities(pkg, synth, small_case):
    """saveKeyFramesAndFactor MO:2136-2142 after the chain: the staged float4 records carry the intensity at byte 12, a packed
    20-byte x,y,z,i@12 PointCloud2 at byte 12 too, PCL records at byte 16 -- the store takes the offset of the upload, it
    does not guess byte 16 (round-2 advisor finding)."""
    sc, pose = _raw_sweep(synth, seed=304, k=1)
    guess = pose.astype(np.float32)
    xyzi = np.concatenate([sc["xyz"], sc["intensity"][:, None]], 1).astype(np.float32)
    ds, _ = pkg.voxel_grid(xyzi, 0.4)
    ident = np.zeros((1, 6), np.float32)
    maps = []
    for flavour in ("host", "chain", "packed20"):
        h = pkg.ScanToMap(); h.set_map(small_case["map"])
        store = pkg.KeyframeStore()
        if flavour == "host":
            kid = store.add(ds)
        elif flavour == "chain":
            lay = pkg.PC2Layout(point_step=32, off_x=0, off_intensity=16, off_ring=-1, off_time=-1)
            h.downsampleAndScan2MapOptimization(_pcl(sc["xyz"], sc["intensity"]), len(xyzi), lay, 0.4, guess)
            kid = store.add_from_handle(h, 0)
        else:
            blob = np.zeros((len(ds), 5), np.float32)
            blob[:, :3], blob[:, 3], blob[:, 4] = ds[:, :3], ds[:, 3], 777.0          # x,y,z,intensity@12, junk@16
            lay = pkg.PC2Layout(point_step=20, off_x=0, off_intensity=12, off_ring=-1, off_time=-1)
            h.scan2MapOptimizationPC2(blob.view(np.uint8), len(ds), lay, guess)
            kid = store.add_from_handle(h, 0)
        out, n_out, _ = store.assemble([kid], ident, 0.3, want_output=True, max_out=len(ds) + 16)
        maps.append(out)
        store.close(); h.close()
    assert len(maps[0]) > 100
    np.testing.assert_array_equal(maps[0].view(np.uint32), maps[1].view(np.uint32))
    np.testing.assert_array_equal(maps[0].view(np.uint32), maps[2].view(np.uint32))


def test_wire_offsets_that_would_wrap_are_refused(pkg, small_case):
    """PointCloud2 field offsets come off the wire: `off_x + 12 > point_step` wraps for off_x = 0xfffffff4 and used to pass."""
    q = small_case["queries"][0]
    blob = np.zeros((len(q["scan"]), 8), np.float32)
    blob[:, :3] = q["scan"]
    h = pkg.ScanToMap(); h.set_map(small_case["map"])
    for off_x in (0xFFFFFFF4, 0xFFFFFFFC, 24, 28):
        bad = pkg.PC2Layout(point_step=32, off_x=off_x, off_intensity=-1, off_ring=-1, off_time=-1)
        with pytest.raises(pkg.LioError):
            h.scan2MapOptimizationPC2(blob.view(np.uint8), len(blob), bad, q["pose_init"])
        with pytest.raises(pkg.LioError):
            h.downsampleAndScan2MapOptimization(blob, len(blob), bad, 0.4, q["pose_init"])
    # the deskew entry point validates ring / time offsets the same way (lio_prepare.hip)
    dg = pkg.deskew_default_config(N_SCAN=16)
    imu = pkg.imu_deskew_info(100.0 - 0.011 + np.arange(70) * 0.002, np.zeros((70, 3)), 100.0, 100.1)
    for kw in (dict(off_x=0xFFFFFFF4), dict(off_ring=0x7FFFFFFF), dict(off_time=0x7FFFFFFC), dict(off_intensity=0x7FFFFFFC)):
        f = dict(point_step=32, off_x=0, off_intensity=16, off_ring=20, ring_type=4, off_time=24, time_type=0)
        f.update(kw)
        with pytest.raises(pkg.LioError):
            pkg.deskew_pc2(dg, blob.view(np.uint8), len(blob), pkg.PC2Layout(**f), 100.0, imu)
    bad_i = pkg.PC2Layout(point_step=32, off_x=0, off_intensity=0x7FFFFFFC, off_ring=-1, off_time=-1)
    with pytest.raises(pkg.LioError):
        h.downsampleAndScan2MapOptimization(blob, len(blob), bad_i, 0.4, q["pose_init"])
    h.close()
