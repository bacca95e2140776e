"""EXTENSION beyond this reference (north_star's `cornerOptimization`): the HIP
point-to-line path through the C ABI vs the CPU restatement (oracle/lio_oracle.c lo_scan2map_cs).
Same bar as the surf path -- association bit-exact, pose within 1e-5 m / 1e-6 rad -- but the oracle
itself has no reference fixture to be pinned against: parity unpinned."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
synth = importlib.import_module("lio-slam_amd.synth")


@pytest.fixture(scope="module")
def corner_case():
    return synth.add_corners(synth.make_case("vlp16", n_keyframes=8, seed=9, device="cpu", n_queries=4), "vlp16", seed=9)


def _check(pose, res, corr_c, corr_s, oracle, cfg, q, case, corr_iter=0):
    po, ro, matP, cc = oracle.scan2map_cs(cfg, q["corners"], case["corner_map"], q["scan"], case["map"],
                                          q["pose_init"], corr_iter=corr_iter)
    assert res.iters == ro.iters and res.converged == ro.converged and res.is_degenerate == ro.is_degenerate
    assert list(res.n_corr_iter)[:res.iters] == list(ro.n_corr_iter)[:ro.iters]
    if corr_c is not None:
        flag, coeff, nn = corr_c
        assert np.array_equal(flag, cc[0]), f"{int((flag != cc[0]).sum())} corner flags differ"
        assert np.array_equal(nn, cc[2])
        gated = nn[:, 0] >= 0                         # coefficients exist for every point that passed the 5-NN gate
        assert np.array_equal(coeff[gated].view(np.uint32), cc[1][gated].view(np.uint32)), "corner coefficients not bit-exact"
        assert flag.sum() > 100
    assert np.abs(pose[3:] - po[3:]).max() <= 1e-5
    assert np.abs(pose[:3] - po[:3]).max() <= 1e-6
    return po, ro


def test_register_cs_matches_oracle(pkg, oracle, corner_case):
    cfg = oracle.default_config(knn_mode=1, n_threads=8)
    s2m = pkg.ScanToMap(record_corr_iter=0)
    s2m.set_map(corner_case["map"])
    s2m.set_corner_map(corner_case["corner_map"])
    for q in corner_case["queries"][:2]:
        pose, res, rc = s2m.scan2MapOptimizationCS(q["corners"], q["scan"], q["pose_init"])
        assert rc == 0
        po, ro = _check(pose, res, s2m.get_corner_correspondences(0), None, oracle, cfg, q, corner_case)
        # the surf correspondences of the same iteration are still those of the surf-only association
        flag, coeff, nn = s2m.get_correspondences(0)
        fo, co, no = oracle.surf_optimization(cfg, q["pose_init"], q["scan"], corner_case["map"])
        assert np.array_equal(flag, fo) and np.array_equal(nn, no)
        assert np.abs(pose[3:] - q["pose_true"][3:]).max() < 0.05
        # normal equations of the last iteration: fp64 sums of exact fp32 products on both sides
        np.testing.assert_allclose(np.frombuffer(res.AtA, np.float32), np.frombuffer(ro.AtA, np.float32), rtol=1e-6)
    s2m.close()


def test_later_iteration_association_bit_exact(pkg, oracle, corner_case):
    cfg = oracle.default_config(knn_mode=1, n_threads=8)
    q = corner_case["queries"][2]
    s2m = pkg.ScanToMap(record_corr_iter=2)
    s2m.set_map(corner_case["map"])
    s2m.set_corner_map(corner_case["corner_map"])
    pose, res, rc = s2m.scan2MapOptimizationCS(q["corners"], q["scan"], q["pose_init"])
    _check(pose, res, s2m.get_corner_correspondences(0), None, oracle, cfg, q, corner_case, corr_iter=2)
    s2m.close()


@pytest.mark.parametrize("graph", [0, 1])
def test_batch_with_and_without_corners(pkg, oracle, corner_case, graph):
    cfg = oracle.default_config(knn_mode=1, n_threads=8)
    qs = corner_case["queries"]
    empty = np.zeros((0, 3), np.float32)
    corners = [qs[0]["corners"], empty, qs[2]["corners"][:300], qs[3]["corners"]]
    s2m = pkg.ScanToMap(use_graph=graph, graph_iters=3)
    s2m.set_map(corner_case["map"])
    s2m.set_corner_map(corner_case["corner_map"])
    for rep in range(2):                                  # second pass re-uses buffers and the cached graph
        s2m.batch_upload([q["scan"] for q in qs])
        s2m.batch_upload_corners(corners)
        s2m.batch_set_poses(np.stack([q["pose_init"] for q in qs]))
        s2m.batch_run()
        poses, res = s2m.batch_results()
        for i, q in enumerate(qs):
            po, ro, _, _ = oracle.scan2map_cs(cfg, corners[i], corner_case["corner_map"], q["scan"], corner_case["map"], q["pose_init"])
            assert res[i].iters == ro.iters
            assert list(res[i].n_corr_iter)[:ro.iters] == list(ro.n_corr_iter)[:ro.iters]
            assert np.abs(poses[i][3:] - po[3:]).max() <= 1e-5 and np.abs(poses[i][:3] - po[:3]).max() <= 1e-6
    # a surf batch uploaded afterwards runs surf-only again (the corner batch does not linger)
    s2m.batch_upload([q["scan"] for q in qs])
    s2m.batch_set_poses(np.stack([q["pose_init"] for q in qs]))
    s2m.batch_run()
    poses, res = s2m.batch_results()
    for i, q in enumerate(qs):
        po, ro = oracle.scan2map(cfg, q["scan"], corner_case["map"], q["pose_init"])[:2]
        assert res[i].iters == ro.iters and np.abs(poses[i] - po).max() <= 1e-5
    s2m.close()


def test_scan_sharded_corner_sums(pkg, oracle, corner_case):
    """Two scan-range shards in one process: partial sums added on the host, as the all-reduce would."""
    import torch
    qs = corner_case["queries"]
    scans = [q["scan"] for q in qs]
    corners = [q["corners"] for q in qs]
    poses0 = np.stack([q["pose_init"] for q in qs])
    ref = pkg.ScanToMap()
    ref.set_map(corner_case["map"]); ref.set_corner_map(corner_case["corner_map"])
    ref.batch_upload(scans); ref.batch_upload_corners(corners); ref.batch_set_poses(poses0); ref.batch_run()
    rp, rr = ref.batch_results()
    ref.close()
    hs, sums = [], []
    for r in range(2):
        h = pkg.ScanToMap()
        h.set_map(corner_case["map"]); h.set_corner_map(corner_case["corner_map"])
        h.set_scan_shard(r, 2)
        h.batch_upload(scans); h.batch_upload_corners(corners); h.batch_set_poses(poses0)
        h.batch_begin()
        hs.append(h)
        sums.append(torch.zeros((len(scans), 32), dtype=torch.float64, device="cuda"))
    for it in range(30):
        for h, s in zip(hs, sums):
            h.batch_iter_partial(s.data_ptr())
        for h in hs:
            h.batch_sync()
        total = sums[0] + sums[1]
        torch.cuda.synchronize()
        for h in hs:
            h.batch_iter_apply(total.data_ptr())
        if hs[0].batch_n_active() == 0:
            break
    for h in hs:
        p, r = h.batch_results()
        assert [x.iters for x in r] == [x.iters for x in rr]
        assert [x.n_corr_last for x in r] == [x.n_corr_last for x in rr]
        np.testing.assert_allclose(p[:, 3:], rp[:, 3:], atol=1e-5)
        np.testing.assert_allclose(p[:, :3], rp[:, :3], atol=1e-6)
        h.close()


def test_corner_api_errors(pkg, corner_case):
    q = corner_case["queries"][0]
    s2m = pkg.ScanToMap()
    s2m.set_map(corner_case["map"])
    with pytest.raises(pkg.LioError):
        s2m.scan2MapOptimizationCS(q["corners"], q["scan"], q["pose_init"])     # no corner map
    s2m.set_corner_map(corner_case["corner_map"])
    s2m.batch_upload([q["scan"]])
    with pytest.raises(pkg.LioError):
        s2m.batch_upload_corners([q["corners"], q["corners"]])                  # scan count mismatch
    # an empty corner map is legal: no edge point finds five neighbours, the loop is surf-only
    s2m.set_corner_map(np.zeros((0, 3), np.float32))
    pose, res, rc = s2m.scan2MapOptimizationCS(q["corners"], q["scan"], q["pose_init"])
    p0, r0, _ = s2m.scan2MapOptimization(q["scan"], q["pose_init"])
    assert np.array_equal(pose, p0) and res.iters == r0.iters
    s2m.close()
