"""Randomised parity: the HIP association against the CPU oracle on clouds that are NOT
lidar scenes -- uniform noise, tight clusters, exact lattices, collinear neighbours, large
coordinates, points outside the map -- where threshold decisions and pivoting paths differ from
the street scenes.  One GN iteration, correspondences recorded: flags, 5-NN sets and plane
coefficients must be bit-exact (NaN patterns included), the normal equations equal after rounding."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _planes(rng, n_planes, pts_per_plane, extent, noise):
    out = []
    for _ in range(n_planes):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        c = rng.uniform(-extent, extent, 3)
        basis = np.linalg.svd(n[None, :])[2][1:]
        uv = rng.uniform(-6, 6, (pts_per_plane, 2))
        out.append(c + uv @ basis + rng.normal(0, noise, (pts_per_plane, 3)))
    return np.concatenate(out)


def _check(pkg, oracle, scan, map_xyz, pose, **cfg):
    s2m = pkg.ScanToMap(record_corr_iter=0, max_iters=1, **cfg)
    s2m.set_map(map_xyz)
    p, res, rc = s2m.scan2MapOptimization(scan, pose)
    flag, coeff, nn = s2m.get_correspondences(0)
    s2m.close()
    ocfg = oracle.default_config(knn_mode=1, n_threads=8, max_iters=1)
    po, ro, _, corr = oracle.scan2map(ocfg, scan, map_xyz, pose, corr_iter=0)
    assert rc == ro.status and res.iters == ro.iters
    np.testing.assert_array_equal(flag, corr[0])
    np.testing.assert_array_equal(nn, corr[2])
    np.testing.assert_array_equal(coeff.view(np.uint32)[flag == 1], corr[1].view(np.uint32)[flag == 1])
    assert res.n_corr_last == ro.n_corr_last
    if ro.n_corr_last >= 50:
        a, b = np.array(res.AtA, np.float32), np.array(ro.AtA, np.float32)
        assert (a.view(np.uint32) != b.view(np.uint32)).sum() <= 2
        np.testing.assert_allclose(p, po, rtol=0, atol=1e-5)
    else:
        np.testing.assert_array_equal(p, po)
    return int(flag.sum()), int((nn[:, 4] >= 0).sum())


@pytest.mark.parametrize("seed", range(6))
def test_random_plane_soup(pkg, oracle, seed):
    rng = np.random.default_rng(seed)
    map_xyz = _planes(rng, 25, 1500, 25.0, 0.02).astype(np.float32)
    scan = (_planes(rng, 25, 300, 25.0, 0.05) ).astype(np.float32)
    scan = np.concatenate([scan, map_xyz[rng.choice(len(map_xyz), 2000)] + rng.normal(0, 0.05, (2000, 3)).astype(np.float32)])
    pose = np.concatenate([rng.normal(0, 0.02, 3), rng.normal(0, 0.1, 3)]).astype(np.float32)
    n_acc, n_gate = _check(pkg, oracle, scan, map_xyz, pose, cell_div=1 + seed % 3, x_sub=(1, 4, 2, 8, 4, 1)[seed], tight_rows=(1, 1, -1, 1, -1, 1)[seed])
    assert n_gate > 500


@pytest.mark.parametrize("seed,cfg", [(20, dict(x_sub=4, tight_rows=1)), (21, dict(x_sub=8, tight_rows=1, cell_div=3)),
                                      (22, dict(x_sub=1, tight_rows=1, pipeline=4)), (23, dict(x_sub=2, tight_rows=-1))])
def test_later_iterations_on_plane_soup(pkg, oracle, seed, cfg):
    """The search bound of the previous iteration picks the row table and the x range of a query from the second iteration
    on (tight rows, finer x buckets): the correspondences of iteration 2 against the oracle's exact search, bit for bit,
    on clouds with a wide spread of neighbour distances (so that both row tables are in use) and a pose that moves."""
    rng = np.random.default_rng(seed)
    map_xyz = _planes(rng, 20, 1200, 20.0, 0.02).astype(np.float32)
    map_xyz = np.concatenate([map_xyz, rng.uniform(-20, 20, (3000, 3)).astype(np.float32)])          # sparse clutter: large d5
    scan = np.concatenate([map_xyz[rng.choice(len(map_xyz), 3000)] + rng.normal(0, 0.04, (3000, 3)).astype(np.float32),
                           _planes(rng, 10, 150, 20.0, 0.05).astype(np.float32)]).astype(np.float32)
    pose = np.concatenate([rng.normal(0, 0.01, 3), rng.normal(0, 0.15, 3)]).astype(np.float32)
    s2m = pkg.ScanToMap(record_corr_iter=2, max_iters=3, **cfg)
    s2m.set_map(map_xyz)
    p, res, rc = s2m.scan2MapOptimization(scan, pose)
    flag, coeff, nn = s2m.get_correspondences(0)
    s2m.close()
    ocfg = oracle.default_config(knn_mode=1, n_threads=8, max_iters=3)
    po, ro, _, corr = oracle.scan2map(ocfg, scan, map_xyz, pose, corr_iter=2)
    assert rc == ro.status and res.iters == ro.iters == 3
    np.testing.assert_array_equal(flag, corr[0])
    np.testing.assert_array_equal(nn, corr[2])
    np.testing.assert_array_equal(coeff.view(np.uint32)[flag == 1], corr[1].view(np.uint32)[flag == 1])
    assert list(res.n_corr_iter)[:3] == list(ro.n_corr_iter)[:3]
    assert int(flag.sum()) > 1000
    np.testing.assert_allclose(p, po, rtol=0, atol=1e-5)


def test_uniform_noise_rejects_most_planes(pkg, oracle):
    rng = np.random.default_rng(10)
    map_xyz = rng.uniform(-8, 8, (60000, 3)).astype(np.float32)        # dense enough that the 1 m gate passes
    scan = rng.uniform(-9, 9, (4000, 3)).astype(np.float32)
    n_acc, n_gate = _check(pkg, oracle, scan, map_xyz, np.zeros(6, np.float32))
    assert n_gate > 3000 and n_acc < n_gate                          # gate passes, plane test mostly fails


def test_collinear_and_duplicate_neighbours(pkg, oracle):
    """Rank-deficient 5x3 systems: Eigen's pivot threshold / zero-pivot paths, NaN planes."""
    rng = np.random.default_rng(11)
    t = np.arange(0, 40, 0.2, dtype=np.float32)
    line = np.stack([t, np.zeros_like(t), np.zeros_like(t)], 1)
    dup = np.repeat(rng.uniform(-5, 5, (200, 3)).astype(np.float32) + np.array([0, 10, 0], np.float32), 6, axis=0)
    origin = np.zeros((10, 3), np.float32)                            # exactly at the origin: zero matrix
    map_xyz = np.concatenate([line, dup, origin, _planes(rng, 4, 800, 5.0, 0.01).astype(np.float32)])
    scan = np.concatenate([line[::3] + np.array([0.05, 0.02, 0.01], np.float32), dup[::6] + 0.01,
                           rng.normal(0, 0.1, (50, 3)).astype(np.float32), _planes(rng, 4, 100, 5.0, 0.03).astype(np.float32)])
    _check(pkg, oracle, scan.astype(np.float32), map_xyz, np.zeros(6, np.float32))
    _check(pkg, oracle, scan.astype(np.float32), map_xyz, np.zeros(6, np.float32), x_sub=4, tight_rows=1)


@pytest.mark.parametrize("offset", [(5000.0, -3000.0, 120.0), (-65536.0, 131072.0, 0.0)])
def test_large_coordinates(pkg, oracle, offset):
    """UTM-like offsets: fp32 spacing up to 1.6 cm, cell coordinates far from zero."""
    rng = np.random.default_rng(12)
    off = np.array(offset, np.float32)
    map_xyz = (_planes(rng, 10, 2000, 15.0, 0.02) + off).astype(np.float32)
    scan = _planes(rng, 10, 400, 15.0, 0.04).astype(np.float32)
    scan = np.concatenate([scan, (map_xyz[::7] - off + rng.normal(0, 0.03, (len(map_xyz[::7]), 3))).astype(np.float32)])
    pose = np.array([0.01, -0.02, 0.03, off[0] + 0.05, off[1] - 0.04, off[2] + 0.02], np.float32)
    n_acc, n_gate = _check(pkg, oracle, scan, map_xyz, pose)
    assert n_gate > 300
    # finer x buckets: at 131072 m one fp32 step is 1.6 cm, a fine cell (0.25 m / 4) holds four of them
    assert _check(pkg, oracle, scan, map_xyz, pose, x_sub=8, cell_div=3) == (n_acc, n_gate)
    assert _check(pkg, oracle, scan, map_xyz, pose, x_sub=4, tight_rows=1) == (n_acc, n_gate)


def test_scan_partly_outside_the_map_and_nonfinite_points(pkg, oracle):
    rng = np.random.default_rng(13)
    map_xyz = _planes(rng, 8, 1500, 6.0, 0.02).astype(np.float32)
    scan = np.concatenate([_planes(rng, 8, 200, 6.0, 0.04), rng.uniform(-500, 500, (500, 3)),
                           rng.uniform(1e6, 1e7, (20, 3))]).astype(np.float32)
    _check(pkg, oracle, scan, map_xyz, np.zeros(6, np.float32))
    _check(pkg, oracle, scan, map_xyz, np.zeros(6, np.float32), x_sub=4, tight_rows=1)
    # non-finite map points are ignored by both sides (they can never be within 1 m of anything)
    bad = map_xyz.copy()
    bad[::50] = np.nan
    bad[1::50, 0] = np.inf
    s2m = pkg.ScanToMap(record_corr_iter=0, max_iters=1)
    s2m.set_map(bad)
    s2m.scan2MapOptimization(scan[:1600], np.zeros(6, np.float32))
    flag, coeff, nn = s2m.get_correspondences(0)
    s2m.close()
    good = np.isfinite(bad).all(1)
    ocfg = oracle.default_config(knn_mode=0, max_iters=1)
    _, _, _, corr = oracle.scan2map(ocfg, scan[:1600], bad[good], np.zeros(6, np.float32), corr_iter=0)
    remap = np.nonzero(good)[0]
    nn_o = np.where(corr[2] >= 0, remap[np.clip(corr[2], 0, None)], -1)
    np.testing.assert_array_equal(flag, corr[0])
    np.testing.assert_array_equal(nn, nn_o)
