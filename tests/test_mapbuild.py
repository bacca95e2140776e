"""Local-map assembly: transformPointCloud MO:849-868 and the
pcl::VoxelGrid centroid filter (MO:1605-1611, MO:1581-1583)."""
import numpy as np
import pytest


def _cloud(rng, n, extent=(40, 30, 6)):
    xyz = rng.uniform(-1, 1, (n, 3)) * np.array(extent)
    return np.concatenate([xyz, rng.uniform(0, 255, (n, 1))], 1).astype(np.float32)


# ------------------------------------------------------------------ oracle (CPU)
def test_oracle_voxel_grid_properties(oracle, synth):
    rng = np.random.default_rng(0)
    pts = _cloud(rng, 20000)
    out, rc = oracle.voxel_grid(pts, 0.4)
    assert rc == 0 and 0 < len(out) < len(pts)
    # same voxelisation and output order as the independent numpy statement in the generator
    ref = synth.voxel_downsample(pts[:, :3], 0.4)
    assert len(ref) == len(out)
    np.testing.assert_allclose(out[:, :3], ref, atol=2e-5)
    # every centroid lies inside its voxel and voxels are unique and ascending (x fastest)
    inv = np.float32(1.0) / np.float32(0.4)
    ijk = np.floor(out[:, :3] * inv).astype(np.int64)
    mn = np.floor(pts[:, :3].min(0) * inv).astype(np.int64)
    div = np.floor(pts[:, :3].max(0) * inv).astype(np.int64) - mn + 1
    key = (ijk[:, 0] - mn[0]) + (ijk[:, 1] - mn[1]) * div[0] + (ijk[:, 2] - mn[2]) * div[0] * div[1]
    assert (np.diff(key) > 0).all()
    # mass is conserved: count-weighted centroids reproduce the mean (checksum of checksums)
    in_key = np.floor(pts[:, :3] * inv).astype(np.int64)
    in_key = (in_key[:, 0] - mn[0]) + (in_key[:, 1] - mn[1]) * div[0] + (in_key[:, 2] - mn[2]) * div[0] * div[1]
    cnt = np.bincount(np.searchsorted(key, in_key), minlength=len(out))
    np.testing.assert_allclose((out.astype(np.float64) * cnt[:, None]).sum(0) / len(pts), pts.astype(np.float64).mean(0), atol=1e-4)
    # a single point, and points all in one voxel
    one, _ = oracle.voxel_grid(pts[:1], 0.4)
    np.testing.assert_array_equal(one, pts[:1])
    same, _ = oracle.voxel_grid(np.tile(pts[:1], (7, 1)), 0.4)
    assert len(same) == 1
    empty, _ = oracle.voxel_grid(pts[:0], 0.4)
    assert len(empty) == 0


def test_oracle_voxel_grid_overflow_passes_through(oracle):
    # leaf 0.01 over a 200 m extent: (20001)^3 voxels > INT32_MAX -> PCL warns and copies the input
    # (the vehicle configs set mappingSurfLeafSize 0.01)
    rng = np.random.default_rng(1)
    pts = _cloud(rng, 500, extent=(100, 100, 100))
    out, rc = oracle.voxel_grid(pts, 0.01)
    assert rc == 1
    np.testing.assert_array_equal(out, pts)


def test_oracle_transform_point_cloud(oracle, synth):
    rng = np.random.default_rng(2)
    pts = _cloud(rng, 1000)
    pose = np.array([0.02, -0.03, 1.2, 10.0, -4.0, 1.8], np.float32)
    out = oracle.transform_point_cloud(pts, pose)
    T = synth.pose_matrix(pose.astype(np.float64))
    ref = pts[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    np.testing.assert_allclose(out[:, :3], ref, atol=2e-5)
    np.testing.assert_array_equal(out[:, 3], pts[:, 3])


# -------------------------------------------------------------------- GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("n,leaf", [(1, 0.4), (63, 0.4), (5000, 0.4), (60000, 0.2), (60000, 2.0), (3000, 25.0), (300000, 0.3), (1100000, 0.5)])
def test_gpu_voxel_grid_bit_exact(pkg, oracle, n, leaf):
    """K7: stable radix sort of (voxel key, input index) pairs + in-order sums (lio_voxsort.h), 1 point .. 1.1 M points (both tile sizes)."""
    pts = _cloud(np.random.default_rng(n), n)
    out_g, rc_g = pkg.voxel_grid(pts, leaf)
    out_o, rc_o = oracle.voxel_grid(pts, leaf)
    assert rc_g == rc_o == 0 and len(out_g) == len(out_o)
    np.testing.assert_array_equal(out_g.view(np.uint32), out_o.view(np.uint32))


@pytest.mark.gpu
def test_gpu_voxel_grid_key_spaces_beyond_the_counting_sort(pkg, oracle):
    """A raw sweep at the reference's small scan leaves spans 10^8..10^9 voxels (mappingSurfLeafSize 0.2 / 0.15, jeep.yaml:99,
    lio_sam_livox.yaml:56): K7 sorts the points instead of keeping arrays over the key space, so everything PCL filters (up
    to 2^31 - 1 voxels) is filtered -- the counting sort of rounds 1-2 refused more than 2^29 and memset gigabytes below that."""
    rng = np.random.default_rng(17)
    pts = _cloud(rng, 120000, extent=(100, 100, 12))
    for leaf in (0.2, 0.15, 0.09):                                   # 1.2e8, 2.8e8 and 1.3e9 voxels
        out_g, rc_g = pkg.voxel_grid(pts, leaf)
        out_o, rc_o = oracle.voxel_grid(pts, leaf)
        assert rc_g == rc_o == 0 and len(out_g) == len(out_o) > 50000
        np.testing.assert_array_equal(out_g.view(np.uint32), out_o.view(np.uint32))


@pytest.mark.gpu
def test_gpu_voxel_grid_edge_cases(pkg, oracle):
    rng = np.random.default_rng(3)
    # many points in very few voxels (> 512 per voxel: the oversized-voxel path)
    dense = (_cloud(rng, 5000, extent=(0.3, 0.3, 0.1)))
    out_g, _ = pkg.voxel_grid(dense, 0.5)
    out_o, _ = oracle.voxel_grid(dense, 0.5)
    np.testing.assert_array_equal(out_g.view(np.uint32), out_o.view(np.uint32))
    assert len(out_g) <= 8
    # both tiers of the centroid pass in one cloud: voxels with 1..128 points (one thread each) and crowded ones
    # (one workgroup, 1024 points at a time through LDS, four component lanes)
    tiers = [_cloud(rng, 40000, extent=(20, 20, 2))]
    for k, cnt in enumerate((60, 128, 129, 700, 4096, 4097, 9000)):
        blob = _cloud(rng, cnt, extent=(0.3, 0.3, 0.3))
        blob[:, :3] = blob[:, :3] * 0.3 + np.array([40.2 + 2.0 * k, 0.2, 0.2], np.float32)   # inside one 0.5 m voxel
        tiers.append(blob)
    mixed = np.concatenate(tiers)
    mixed = mixed[rng.permutation(len(mixed))]
    out_g, _ = pkg.voxel_grid(mixed, 0.5)
    out_o, _ = oracle.voxel_grid(mixed, 0.5)
    np.testing.assert_array_equal(out_g.view(np.uint32), out_o.view(np.uint32))
    # overflow: pass-through, status 1
    big = _cloud(rng, 500, extent=(100, 100, 100))
    out_g, rc = pkg.voxel_grid(big, 0.01)
    assert rc == 1
    np.testing.assert_array_equal(out_g, big)
    out_g, rc = pkg.voxel_grid(big[:0], 0.4)
    assert rc == 0 and len(out_g) == 0


@pytest.mark.gpu
def test_gpu_assemble_map_matches_oracle_and_feeds_registration(pkg, oracle, synth):
    rng = np.random.default_rng(4)
    boxes = synth.make_scene(31, length=60.0)
    kfs = synth.keyframe_poses(6, seed=31)
    clouds, poses = [], []
    for k, kp in enumerate(kfs):
        sc = synth.cast_scan(boxes, kp, "vlp16", seed=100 + k, device="cpu")
        xyz, _ = oracle.voxel_grid(np.concatenate([sc["xyz"], sc["intensity"][:, None]], 1), 0.4)
        clouds.append(xyz)
        poses.append(np.asarray(kp, np.float32))
    poses = np.stack(poses)
    # oracle: extractCloud = concat of transformed keyframes -> VoxelGrid(0.5)   (MO:1556-1588)
    world = np.concatenate([oracle.transform_point_cloud(c, p) for c, p in zip(clouds, poses)])
    map_o, _ = oracle.voxel_grid(world, 0.5)
    s2m = pkg.ScanToMap()
    map_g, n_out, rc = pkg.assemble_map(clouds, poses, 0.5, s2m=s2m)
    assert rc == 0 and n_out == len(map_o)
    np.testing.assert_array_equal(map_g.view(np.uint32), map_o.view(np.uint32))
    # the map installed on the device by assemble_map registers exactly like set_map(map)
    tp = np.array(kfs[-1]); tp[3] += 0.5
    scan, init = synth.make_query(boxes, tp, "vlp16", seed=777, device="cpu")
    pose_a, res_a, _ = s2m.scan2MapOptimization(scan, init)
    ref = pkg.ScanToMap()
    ref.set_map(map_o[:, :3])
    pose_b, res_b, _ = ref.scan2MapOptimization(scan, init)
    np.testing.assert_array_equal(pose_a, pose_b)
    assert res_a.iters == res_b.iters and res_a.converged == 1
    assert np.abs(pose_a[3:] - tp[3:]).max() < 0.05
    # no keyframes: empty map
    out, n0, rc = pkg.assemble_map([], np.zeros((0, 6), np.float32), 0.5)
    assert n0 == 0
    # resident keyframe store (surfCloudKeyFrames kept in HBM): a subset, in another order, new poses
    store = pkg.KeyframeStore()
    ids = [store.add(c) for c in clouds]
    assert ids == list(range(len(clouds))) and len(store) == len(clouds)
    sel = [4, 1, 3]
    poses2 = poses[sel] + np.array([0.001, -0.002, 0.01, 0.05, -0.03, 0.0], np.float32)
    world2 = np.concatenate([oracle.transform_point_cloud(clouds[i], p) for i, p in zip(sel, poses2)])
    map_o2, _ = oracle.voxel_grid(world2, 0.5)
    map_g2, n2, rc = store.assemble(sel, poses2, 0.5, s2m=s2m, max_out=len(world2))
    assert n2 == len(map_o2)
    np.testing.assert_array_equal(map_g2.view(np.uint32), map_o2.view(np.uint32))
    pose_c, res_c, _ = s2m.scan2MapOptimization(scan, init)          # registers against the new resident map
    ref.set_map(map_o2[:, :3])
    pose_d, res_d, _ = ref.scan2MapOptimization(scan, init)
    np.testing.assert_array_equal(pose_c, pose_d)
    with pytest.raises(pkg.LioError):
        store.assemble([99], poses2[:1], 0.5)
    store.close(); s2m.close(); ref.close()


@pytest.mark.gpu
def test_gpu_voxel_grid_fuzz(pkg, oracle):
    """Seeded random clouds: sizes across both sort tiles (4 / 8 items per thread), leaves from a few centimetres to metres,
    extents from one voxel to a 2^30-voxel box, clusters that crowd single voxels -- bit-exact against the oracle every time."""
    rng = np.random.default_rng(20241022)
    for trial in range(14):
        n = int(rng.choice([1, 2, 65, 1000, 4097, 30000, 70000, 270000]))
        ext = (float(rng.choice([0.05, 3.0, 40.0, 120.0])), float(rng.choice([0.05, 20.0, 120.0])), float(rng.choice([0.01, 4.0, 15.0])))
        leaf = float(rng.choice([0.07, 0.15, 0.4, 1.3]))
        pts = _cloud(rng, n, extent=ext)
        if trial % 3 == 0 and n > 100:                       # a third of the points piled into a few spots
            k = n // 3
            pts[:k, :3] = pts[rng.integers(0, n, 4)][:, :3][rng.integers(0, 4, k)] + rng.normal(0, 0.02, (k, 3)).astype(np.float32)
            pts = pts[rng.permutation(n)]
        out_g, rc_g = pkg.voxel_grid(pts, leaf)
        out_o, rc_o = oracle.voxel_grid(pts, leaf)
        assert rc_g == rc_o and len(out_g) == len(out_o), (trial, n, ext, leaf)
        np.testing.assert_array_equal(out_g.view(np.uint32), out_o.view(np.uint32), err_msg=str((trial, n, ext, leaf)))
