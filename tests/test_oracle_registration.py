"""Known-answer tests of the oracle's registration loop on analytic scenes
(pattern of the reference's grid_map_pcl tests: seeded synthetic scene ->
pipeline -> compare with the analytic answer)."""
import numpy as np
import pytest


def test_recovers_true_pose_street(oracle, small_case):
    cfg = oracle.default_config(knn_mode=1, n_threads=4)
    for q in small_case["queries"]:
        pose, res, matP, _ = oracle.scan2map(cfg, q["scan"], small_case["map"], q["pose_init"])
        assert res.status == 0 and res.converged == 1 and res.is_degenerate == 0
        assert 2 <= res.iters <= 15
        assert np.abs(pose[3:] - q["pose_true"][3:]).max() < 0.05, (pose, q["pose_true"])
        assert np.abs(pose[:3] - q["pose_true"][:3]).max() < 0.01
        # error shrinks relative to the initial guess
        assert np.linalg.norm(pose[3:] - q["pose_true"][3:]) < np.linalg.norm(q["pose_init"][3:] - q["pose_true"][3:])


def test_brute_force_and_kdtree_give_identical_registration(oracle, small_case):
    q = small_case["queries"][0]
    a = oracle.scan2map(oracle.default_config(knn_mode=0), q["scan"][::3], small_case["map"], q["pose_init"], corr_iter=0)
    b = oracle.scan2map(oracle.default_config(knn_mode=1), q["scan"][::3], small_case["map"], q["pose_init"], corr_iter=0)
    np.testing.assert_array_equal(a[0], b[0])
    for x, y in zip(a[3], b[3]):
        np.testing.assert_array_equal(x, y)


def test_openmp_threads_do_not_change_results(oracle, small_case):
    q = small_case["queries"][1]
    a = oracle.scan2map(oracle.default_config(n_threads=1), q["scan"], small_case["map"], q["pose_init"])
    b = oracle.scan2map(oracle.default_config(n_threads=8), q["scan"], small_case["map"], q["pose_init"])
    np.testing.assert_array_equal(a[0], b[0])
    assert a[1].iters == b[1].iters


def test_exact_jacobian_also_converges(oracle, small_case):
    q = small_case["queries"][0]
    pose, res, _, _ = oracle.scan2map(oracle.default_config(jacobian_mode=1), q["scan"], small_case["map"], q["pose_init"])
    assert res.converged == 1
    assert np.abs(pose[3:] - q["pose_true"][3:]).max() < 0.05


def test_corridor_is_degenerate(oracle, synth):
    """Ground + two parallel walls: translation along x is unobservable, the
    eigenvalue test (MO:1786-1808) must flag it and matP must project it out."""
    case = synth.make_case("vlp16", n_keyframes=5, seed=3, kind="corridor", device="cpu")
    q = case["queries"][0]
    pose, res, matP, _ = oracle.scan2map(oracle.default_config(), q["scan"], case["map"], q["pose_init"])
    assert res.is_degenerate == 1
    assert np.linalg.matrix_rank(matP.astype(np.float64), tol=1e-3) < 6
    np.testing.assert_allclose(matP @ matP, matP, atol=2e-3)      # a projector
    # the unobservable direction keeps (almost) the initial guess: x-translation update is projected away
    assert abs(pose[3] - q["pose_init"][3]) < 0.02
    # the observable ones are still corrected
    assert abs(pose[4] - q["pose_true"][4]) < 0.05 and abs(pose[5] - q["pose_true"][5]) < 0.05


def test_guards(oracle, small_case):
    q = small_case["queries"][0]
    cfg = oracle.default_config()
    pose, res, _, _ = oracle.scan2map(cfg, q["scan"][:30], small_case["map"], q["pose_init"])
    assert res.status == 1 and res.iters == 0 and np.array_equal(pose, q["pose_init"])   # MO:1844
    pose, res, _, _ = oracle.scan2map(cfg, q["scan"][:31], small_case["map"], q["pose_init"])
    assert res.status == 2 and res.iters == 30 and np.array_equal(pose, q["pose_init"])  # < 50 corr, MO:1721
    # matP / isDegenerate carry over untouched when iteration 0 has too few correspondences (MO:177)
    P0 = np.arange(36, dtype=np.float32).reshape(6, 6)
    pose, res, matP, _ = oracle.scan2map(cfg, q["scan"][:31], small_case["map"], q["pose_init"], matP=P0, is_degenerate=1)
    np.testing.assert_array_equal(matP, P0)
    assert res.is_degenerate == 1


def test_force_all_iters(oracle, small_case):
    q = small_case["queries"][0]
    pose, res, _, _ = oracle.scan2map(oracle.default_config(force_all_iters=1, max_iters=9), q["scan"], small_case["map"], q["pose_init"])
    assert res.iters == 9 and res.converged == 1


def test_config0_vlp16_scan_vs_one_keyframe_map(oracle, synth):
    """BASELINE.json configs[0]: a single VLP-16 16x1800 scan against a ONE-keyframe local map -- the second scan a node ever
    sees (the guards MO:1841-1846 pass for the first time: one keyframe exists, the map is that keyframe's cloud through
    VoxelGrid 0.4 then 0.5).  CPU plumbing config: the oracle recovers the pose, brute force and kd-tree agree."""
    case = synth.make_case("vlp16", n_keyframes=1, seed=20241022, device="cpu", n_queries=2)
    assert 2000 < len(case["map"]) < 9000                        # N_m ~ 4-8 k for one keyframe
    for q in case["queries"]:
        assert 2000 < len(q["scan"]) < 9000
        a = oracle.scan2map(oracle.default_config(knn_mode=1, n_threads=4), q["scan"], case["map"], q["pose_init"], corr_iter=0)
        b = oracle.scan2map(oracle.default_config(knn_mode=0, n_threads=4), q["scan"], case["map"], q["pose_init"], corr_iter=0)
        np.testing.assert_array_equal(a[0], b[0])
        for x, y in zip(a[3], b[3]):
            np.testing.assert_array_equal(x, y)
        pose, res = a[0], a[1]
        assert res.status == 0 and res.converged == 1 and res.is_degenerate == 0 and res.n_corr_last > 500
        # the map is one sparse keyframe 0.5 m behind the query: a coarser answer than against 50 keyframes, still centimetres
        assert np.abs(pose[3:] - q["pose_true"][3:]).max() < 0.08, (pose, q["pose_true"])
        assert np.abs(pose[:3] - q["pose_true"][:3]).max() < 0.05        # (pitch is weakly constrained by one VLP-16 keyframe)
