"""EXTENSION beyond this reference: point-to-line residuals of upstream LIO-SAM's
cornerOptimization.  No reference fixture exists for it (parity unpinned); these are known-answer
and self-consistency checks of the CPU restatement in oracle/lio_oracle.c."""
import importlib

import numpy as np
import pytest

synth = importlib.import_module("lio-slam_amd.synth")


def test_eigen3_against_lapack(oracle):
    rng = np.random.default_rng(5)
    for _ in range(300):
        J = rng.normal(size=(5, 3)) * rng.uniform(0.01, 3.0)
        A = (J.T @ J / 5).astype(np.float32)
        w, v = oracle.eigen3(A)
        wr = np.linalg.eigvalsh(A.astype(np.float64))[::-1]
        scale = np.abs(wr).max()
        assert np.all(np.diff(w) <= 0)                                       # descending
        np.testing.assert_allclose(w, wr, atol=3e-6 * scale + 5e-7)
        for i in range(3):                                                    # rows are eigenvectors
            np.testing.assert_allclose(A.astype(np.float64) @ v[i], w[i] * v[i], atol=3e-5 * scale + 5e-7)   # the sweep stops at |pivot| <= FLT_EPSILON (absolute)
        np.testing.assert_allclose(v @ v.T, np.eye(3), atol=1e-5)


def test_eigen3_diagonal_and_zero(oracle):
    w, v = oracle.eigen3(np.diag([1.0, 3.0, 2.0]))
    assert list(w) == [3.0, 2.0, 1.0]
    assert np.array_equal(np.abs(v), np.eye(3)[[1, 2, 0]])
    w, v = oracle.eigen3(np.zeros((3, 3)))
    assert list(w) == [0.0, 0.0, 0.0] and np.array_equal(v, np.eye(3))


def test_point_to_line_known_answer(oracle):
    # a vertical pole through (3, 2): distance and gradient of the point-to-line residual
    z = np.linspace(0.0, 5.0, 51)
    pole = np.stack([np.full_like(z, 3.0), np.full_like(z, 2.0), z], 1).astype(np.float32)
    q = np.array([[3.2, 2.1, 1.0], [3.0, 2.3, 4.0], [9.0, 9.0, 9.0]], np.float32)
    cfg = oracle.default_config(knn_mode=0)
    flag, coeff, nn = oracle.corner_optimization(cfg, np.zeros(6, np.float32), q, pole)
    assert list(flag) == [1, 1, 0]
    assert np.all(nn[2] == -1)                                                # 5th neighbour farther than 1 m
    for i, (dx, dy) in enumerate([(0.2, 0.1), (0.0, 0.3)]):
        d = np.hypot(dx, dy)
        s = 1 - 0.9 * d
        np.testing.assert_allclose(coeff[i], [s * dx / d, s * dy / d, 0.0, s * d], atol=2e-5)
    # the five neighbours are the pole samples closest in height
    assert set(nn[0]) == {8, 9, 10, 11, 12}


def test_blob_is_not_a_line(oracle):
    rng = np.random.default_rng(2)
    blob = rng.normal(0, 0.2, (200, 3)).astype(np.float32)
    cfg = oracle.default_config(knn_mode=0)
    flag, _, nn = oracle.corner_optimization(cfg, np.zeros(6, np.float32), np.zeros((1, 3), np.float32), blob)
    assert nn[0, 0] >= 0 and flag[0] in (0, 1)
    # an isotropic cluster must fail lambda0 > 3*lambda1 far more often than not
    q = rng.normal(0, 0.1, (100, 3)).astype(np.float32)
    flag, _, _ = oracle.corner_optimization(cfg, np.zeros(6, np.float32), q, blob)
    assert flag.mean() < 0.5


def test_kdtree_and_brute_force_agree(oracle):
    case = synth.add_corners(synth.make_case("vlp16", n_keyframes=5, seed=3, device="cpu"), "vlp16", seed=3)
    q = case["queries"][0]
    a = oracle.corner_optimization(oracle.default_config(knn_mode=0), q["pose_init"], q["corners"], case["corner_map"])
    b = oracle.corner_optimization(oracle.default_config(knn_mode=1), q["pose_init"], q["corners"], case["corner_map"])
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert a[0].sum() > 100


def test_no_corners_is_the_reference_loop(oracle):
    case = synth.make_case("vlp16", n_keyframes=5, seed=3, device="cpu")
    q = case["queries"][0]
    cfg = oracle.default_config(knn_mode=1)
    p0, r0 = oracle.scan2map(cfg, q["scan"], case["map"], q["pose_init"])[:2]
    p1, r1, _, _ = oracle.scan2map_cs(cfg, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32),
                                      q["scan"], case["map"], q["pose_init"])
    assert np.array_equal(p0, p1) and r0.iters == r1.iters
    assert np.array_equal(np.frombuffer(r0.AtA, np.float32), np.frombuffer(r1.AtA, np.float32))


def test_combined_registration_recovers_pose(oracle):
    case = synth.add_corners(synth.make_case("vlp16", n_keyframes=8, seed=9, device="cpu", n_queries=2), "vlp16", seed=9)
    cfg = oracle.default_config(knn_mode=1)
    for q in case["queries"]:
        p, r, _, _ = oracle.scan2map_cs(cfg, q["corners"], case["corner_map"], q["scan"], case["map"], q["pose_init"])
        assert r.status == 0 and r.converged == 1
        assert np.abs(p[3:] - q["pose_true"][3:]).max() < 0.05
        assert np.abs(p[:3] - q["pose_true"][:3]).max() < 0.01
        p_s, r_s = oracle.scan2map(cfg, q["scan"], case["map"], q["pose_init"])[:2]
        assert r.n_corr_iter[0] > r_s.n_corr_iter[0]                          # the edge rows joined the system
