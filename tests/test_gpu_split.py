"""The neighbour certificate -- as the split Gauss-Newton pipeline (cfg.pipeline = 2: certificate / candidate scan / fit as
three launches, lio-slam_amd/csrc/lio_split.hip) and inside the fused launch (cfg.pipeline = 3, k_s2m_iterate_cert,
lio_cert.hip) -- against the plain fused launch (pipeline = 1) and the CPU oracle.

The certificate replaces the candidate scan of MO:1631 by a proof that the five nearest neighbours are
among the eight cached ones; nothing observable may change: every iteration's pose, correspondence
count, normal matrix and the recorded association of ANY iteration are compared bit for bit, on street
scans, on a lattice full of exactly tied distances and on ragged batches replayed from a hipGraph."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(pkg, case_map, scan, pose0, **cfg):
    s2m = pkg.ScanToMap(profile=1, **cfg)
    s2m.set_map(case_map)
    pose, res, rc = s2m.scan2MapOptimization(scan, pose0)
    corr = s2m.get_correspondences(0) if cfg.get("record_corr_iter", -1) >= 0 else None
    prof = s2m.profile()
    s2m.close()
    return pose, res, rc, corr, prof


def _same(a, b):
    (pa, ra, rca, ca, _), (pb, rb, rcb, cb, _) = a, b
    assert rca == rcb and ra.iters == rb.iters and ra.converged == rb.converged and ra.is_degenerate == rb.is_degenerate
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(np.array(ra.pose_iter, np.float32), np.array(rb.pose_iter, np.float32))
    assert list(ra.n_corr_iter) == list(rb.n_corr_iter)
    np.testing.assert_array_equal(np.array(ra.AtA, np.float32).view(np.uint32), np.array(rb.AtA, np.float32).view(np.uint32))
    np.testing.assert_array_equal(np.array(ra.matP, np.float32).view(np.uint32), np.array(rb.matP, np.float32).view(np.uint32))
    if ca is not None:
        for x, y in zip(ca, cb):
            np.testing.assert_array_equal(x.view(np.uint8) if x.dtype == np.uint8 else x.view(np.uint32),
                                          y.view(np.uint8) if y.dtype == np.uint8 else y.view(np.uint32))


@pytest.mark.parametrize("pipe", [2, 3, 5])
@pytest.mark.parametrize("rec", [0, 1, 2, 4, 7])
def test_split_equals_fused_every_iteration(pkg, oracle, small_case, rec, pipe):
    for q in small_case["queries"][:2]:
        kw = dict(record_corr_iter=rec, force_all_iters=1, max_iters=9)
        fused = _run(pkg, small_case["map"], q["scan"], q["pose_init"], pipeline=1, **kw)
        split = _run(pkg, small_case["map"], q["scan"], q["pose_init"], pipeline=pipe, **kw)
        assert fused[4].pipeline == 1 and split[4].pipeline == pipe
        _same(fused, split)
        ocfg = oracle.default_config(knn_mode=1, n_threads=8, force_all_iters=1, max_iters=9)
        _, ro, _, corr = oracle.scan2map(ocfg, q["scan"], small_case["map"], q["pose_init"], corr_iter=rec)
        np.testing.assert_array_equal(split[3][0], corr[0])
        np.testing.assert_array_equal(split[3][2], corr[2])
        assert list(ro.n_corr_iter)[:9] == list(split[1].n_corr_iter)[:9]
        # the certificate must actually fire once the pose settles (otherwise this test proves nothing)
        if pipe == 2:
            cert, scanned = np.array(split[4].cert_points[:9]), np.array(split[4].scan_points[:9])
            assert cert[0] == len(q["scan"]) and 0.9 * cert[0] < scanned[0] <= cert[0]    # (points outside the grid are never scanned)
            assert scanned[5:9].sum() < 0.25 * cert[5:9].sum(), (cert, scanned)


@pytest.mark.parametrize("pipe", [2, 3, 5])
@pytest.mark.parametrize("variant", [dict(cell_div=1), dict(cell_div=3, sort_scan=0), dict(sort_scan=2), dict(cell_size=1.7),
                                     dict(jacobian_mode=1), dict(xcd_remap=0)])
def test_split_variants(pkg, small_case, variant, pipe):
    q = small_case["queries"][2]
    kw = dict(record_corr_iter=3, **variant)
    _same(_run(pkg, small_case["map"], q["scan"], q["pose_init"], pipeline=1, **kw),
          _run(pkg, small_case["map"], q["scan"], q["pose_init"], pipeline=pipe, **kw))


@pytest.mark.parametrize("pipe", [2, 3, 5])
def test_split_on_a_lattice_of_tied_distances(pkg, oracle, pipe):
    """Exactly equal distances everywhere (0.5 m lattice, duplicated points, scan points ON map points) while
    the pose creeps by millimetres: the certificate may only fire when no outsider can TIE with the 5th."""
    g = np.arange(0, 8, 0.5, dtype=np.float32)
    lattice = np.stack(np.meshgrid(g, g, np.array([0.0, 0.5], np.float32), indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(0)
    map_xyz = np.concatenate([lattice, lattice[::5]]).astype(np.float32)
    scan = np.concatenate([lattice[::3] + np.float32(0.25), lattice[1::7],
                           rng.uniform(0.5, 7.5, (600, 3)).astype(np.float32) * np.array([1, 1, 0.1], np.float32)])
    pose0 = np.array([0.0, 0.0, 0.002, 0.004, -0.003, 0.001], np.float32)
    for rec in (0, 1, 3):
        kw = dict(record_corr_iter=rec, force_all_iters=1, max_iters=5)
        fused = _run(pkg, map_xyz, scan, pose0, pipeline=1, **kw)
        split = _run(pkg, map_xyz, scan, pose0, pipeline=pipe, **kw)
        _same(fused, split)
        ocfg = oracle.default_config(knn_mode=1, n_threads=8, force_all_iters=1, max_iters=5)
        _, ro, _, corr = oracle.scan2map(ocfg, scan, map_xyz, pose0, corr_iter=rec)
        np.testing.assert_array_equal(split[3][2], corr[2])
        np.testing.assert_array_equal(split[3][0], corr[0])


@pytest.mark.parametrize("pipe", [2, 3, 5])
@pytest.mark.parametrize("graph_iters", [0, 3, 30])
def test_split_batches_and_hipgraph(pkg, small_case, graph_iters, pipe):
    qs = small_case["queries"]
    scans = [q["scan"] for q in qs] + [qs[0]["scan"][:20], qs[1]["scan"][::2], qs[2]["scan"][:300]]
    poses0 = np.stack([q["pose_init"] for q in qs] + [qs[0]["pose_init"], qs[1]["pose_init"], qs[2]["pose_init"]])
    ref = pkg.ScanToMap(pipeline=1)
    ref.set_map(small_case["map"])
    ref.batch_upload(scans); ref.batch_set_poses(poses0); ref.batch_run()
    pr, rr = ref.batch_results()
    s = pkg.ScanToMap(pipeline=pipe, use_graph=1 if graph_iters else 0, graph_iters=max(graph_iters, 1))
    s.set_map(small_case["map"])
    for rep in range(2):                       # the second round starts from a warm cache and must not use it
        s.batch_upload(scans) if rep == 0 else None
        s.batch_set_poses(poses0); s.batch_run()
        ps, rs = s.batch_results()
        np.testing.assert_array_equal(ps, pr)
        assert [r.iters for r in rs] == [r.iters for r in rr] and [r.status for r in rs] == [r.status for r in rr]
        for a, b in zip(rs, rr):
            np.testing.assert_array_equal(np.array(a.AtA, np.float32).view(np.uint32), np.array(b.AtA, np.float32).view(np.uint32))
    # a new map under the resident batch: cached neighbour indices of the old map must be dropped
    half = small_case["map"][::2]
    s.set_map(half); s.batch_set_poses(poses0); s.batch_run()
    ps2, _ = s.batch_results()
    ref.set_map(half); ref.batch_set_poses(poses0); ref.batch_run()
    pr2, _ = ref.batch_results()
    np.testing.assert_array_equal(ps2, pr2)
    ref.close(); s.close()


@pytest.mark.parametrize("pipe", [2, 3, 5])
@pytest.mark.parametrize("offset", [(5000.0, -3000.0, 120.0), (-65536.0, 131072.0, 0.0)])
def test_split_large_coordinates(pkg, offset, pipe):
    """UTM-like offsets (fp32 spacing up to 1.6 cm): the certificate's margins are relative to distances, not
    to coordinates."""
    from test_gpu_fuzz import _planes
    rng = np.random.default_rng(12)
    off = np.array(offset, np.float32)
    map_xyz = (_planes(rng, 10, 2000, 15.0, 0.02) + off).astype(np.float32)
    scan = np.concatenate([_planes(rng, 10, 400, 15.0, 0.04),
                           (map_xyz[::7] - off + rng.normal(0, 0.03, (len(map_xyz[::7]), 3)))]).astype(np.float32)
    pose = np.array([0.01, -0.02, 0.03, off[0] + 0.05, off[1] - 0.04, off[2] + 0.02], np.float32)
    kw = dict(record_corr_iter=4, force_all_iters=1, max_iters=6)
    _same(_run(pkg, map_xyz, scan, pose, pipeline=1, **kw), _run(pkg, map_xyz, scan, pose, pipeline=pipe, **kw))
