"""BASELINE.json configs at (near) full size on the GPU, through the C ABI, checked by the CPU
oracle where it finishes in seconds and by size-independent properties otherwise."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL_T, TOL_R = 1e-5, 1e-6


@pytest.fixture(scope="module")
def vlp16_50kf(synth):
    """configs[1]: VLP-16 16x1800 scan vs 50-keyframe local map."""
    return synth.make_case("vlp16", n_keyframes=50, seed=20241022, device="cpu", n_queries=4)


def test_config2_vlp16_vs_50_keyframes(pkg, oracle, vlp16_50kf):
    case = vlp16_50kf
    s2m = pkg.ScanToMap(record_corr_iter=0)
    s2m.set_map(case["map"])
    cfg = oracle.default_config(knn_mode=1, n_threads=8)
    for q in case["queries"]:
        pose, res, rc = s2m.scan2MapOptimization(q["scan"], q["pose_init"])
        flag, coeff, nn = s2m.get_correspondences(0)
        pose_o, res_o, _, corr = oracle.scan2map(cfg, q["scan"], case["map"], q["pose_init"], corr_iter=0)
        assert rc == res_o.status == 0 and res.iters == res_o.iters and res.converged == 1
        assert np.array_equal(flag, corr[0]) and np.array_equal(nn, corr[2])
        assert np.array_equal(coeff[flag == 1].view(np.uint32), corr[1][flag == 1].view(np.uint32))
        assert np.abs(pose[3:] - pose_o[3:]).max() <= TOL_T and np.abs(pose[:3] - pose_o[:3]).max() <= TOL_R
        assert np.abs(pose[3:] - q["pose_true"][3:]).max() < 0.05       # known answer
    s2m.close()


def test_config3_os1_64_forced_30_iterations(pkg, oracle, synth):
    """configs[2]: OS1-64 64x1024 scan, 30 Gauss-Newton iterations forced (the map uses 40
    keyframes so that the CPU ray caster stays within test time; bench.py runs 200)."""
    case = synth.make_case("os1_64", n_keyframes=40, seed=5, device="cpu", n_queries=1)
    q = case["queries"][0]
    s2m = pkg.ScanToMap(force_all_iters=1, max_iters=30)
    s2m.set_map(case["map"])
    pose, res, rc = s2m.scan2MapOptimization(q["scan"], q["pose_init"])
    pose_o, res_o, _, _ = oracle.scan2map(oracle.default_config(knn_mode=1, n_threads=8, force_all_iters=1),
                                          q["scan"], case["map"], q["pose_init"])
    assert res.iters == res_o.iters == 30 and res.converged == res_o.converged == 1
    assert list(res.n_corr_iter)[:30] == list(res_o.n_corr_iter)[:30]
    np.testing.assert_allclose(np.array(res.pose_iter)[:30, 3:], np.array(res_o.pose_iter)[:30, 3:], atol=TOL_T)
    np.testing.assert_allclose(np.array(res.pose_iter)[:30, :3], np.array(res_o.pose_iter)[:30, :3], atol=TOL_R)
    s2m.close()


def test_config5_batched_properties(pkg, vlp16_50kf, synth):
    """configs[4] shape (many scans streamed against one resident map): properties that need no oracle."""
    case = vlp16_50kf
    rng = np.random.default_rng(0)
    base = case["queries"]
    scans, poses0 = [], []
    for i in range(48):                       # 48 different initial guesses over 4 scans
        q = base[i % len(base)]
        p = q["pose_true"].astype(np.float64).copy()
        p[3:] += rng.normal(0, 0.08, 3); p[:3] += rng.normal(0, 0.01, 3)
        scans.append(q["scan"]); poses0.append(p.astype(np.float32))
    poses0 = np.stack(poses0)
    s2m = pkg.ScanToMap()
    s2m.set_map(case["map"])
    s2m.batch_upload(scans); s2m.batch_set_poses(poses0); s2m.batch_run()
    poses, res = s2m.batch_results()
    # (1) every scan converges to (nearly) the same pose as its siblings that share the scan
    for k in range(len(base)):
        grp = poses[k::len(base)]
        assert np.abs(grp[:, 3:] - grp[0, 3:]).max() < 5e-3 and np.abs(grp[:, :3] - grp[0, :3]).max() < 1e-3
        assert np.abs(grp[:, 3:] - base[k]["pose_true"][3:]).max() < 0.05
    assert all(r.converged == 1 and r.status == 0 for r in res)
    # (2) permutation invariance: the order of the scans in a batch does not change any result (bit-exact)
    perm = rng.permutation(len(scans))
    s2m.batch_upload([scans[i] for i in perm]); s2m.batch_set_poses(poses0[perm]); s2m.batch_run()
    poses_p, res_p = s2m.batch_results()
    np.testing.assert_array_equal(poses_p, poses[perm])
    assert [r.iters for r in res_p] == [res[i].iters for i in perm]
    # (3) a batch gives what single registrations give
    one = pkg.ScanToMap()
    one.set_map(case["map"])
    for i in (0, 7, 31):
        p1, r1, _ = one.scan2MapOptimization(scans[i], poses0[i])
        np.testing.assert_array_equal(p1, poses[i])
    # (4) idempotence: restarting from the converged pose stops within two iterations and barely moves
    s2m.batch_upload(scans); s2m.batch_set_poses(poses); s2m.batch_run()
    poses2, res2 = s2m.batch_results()
    assert max(r.iters for r in res2) <= 3
    assert np.abs(poses2[:, 3:] - poses[:, 3:]).max() < 2e-3 and np.abs(poses2[:, :3] - poses[:, :3]).max() < 2e-4
    one.close(); s2m.close()


def test_neighbour_cache_is_exact(pkg, oracle, small_case):
    """cfg.nn_cache narrows the candidate run from the second iteration on; nothing observable may change:
    every iteration's pose, correspondence count and the recorded association of a late iteration."""
    import numpy as np
    q = small_case["queries"][0]
    outs = []
    for nn_cache in (0, 1):
        s2m = pkg.ScanToMap(nn_cache=nn_cache, record_corr_iter=3, force_all_iters=1, max_iters=8)
        s2m.set_map(small_case["map"])
        pose, res, rc = s2m.scan2MapOptimization(q["scan"], q["pose_init"])
        outs.append((pose, np.array(res.pose_iter, np.float32), list(res.n_corr_iter), s2m.get_correspondences(0),
                     np.array(res.AtA, np.float32)))
        s2m.close()
    a, b = outs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    for x, y in zip(a[3], b[3]):
        assert np.array_equal(x, y)
    assert np.array_equal(a[4], b[4])
    cfg = oracle.default_config(knn_mode=1, n_threads=8, force_all_iters=1, max_iters=8)
    po, ro, _, corr = oracle.scan2map(cfg, q["scan"], small_case["map"], q["pose_init"], corr_iter=3)
    assert np.array_equal(b[3][0], corr[0]) and np.array_equal(b[3][2], corr[2])
    assert list(ro.n_corr_iter)[:8] == b[2][:8]


def test_persistent_members_survive_pose_only_reads(pkg, synth):
    """isDegenerate / matP are members of the node (MO:176-177): a registration that never reaches the solve
    (< 50 correspondences, MO:1721-1724) leaves them as the previous frame set them -- also when the
    previous results were read back poses-only (the compact summary path)."""
    import numpy as np
    case = synth.make_case("vlp16", n_keyframes=4, seed=22, kind="corridor", device="cpu")
    q = case["queries"][0]
    s2m = pkg.ScanToMap()
    s2m.set_map(case["map"])
    s2m.batch_upload([q["scan"]]); s2m.batch_set_poses(q["pose_init"][None]); s2m.batch_run()
    poses, _ = s2m.batch_results(with_results=False)                 # poses only
    ref = pkg.ScanToMap()
    ref.set_map(case["map"])
    _, res_ref, _ = ref.scan2MapOptimization(q["scan"], q["pose_init"])
    assert res_ref.is_degenerate == 1
    tiny = q["scan"][:40]                                            # > 30 points, < 50 correspondences
    s2m.batch_upload([tiny]); s2m.batch_set_poses(poses); s2m.batch_run()
    _, res = s2m.batch_results(with_results=True)
    assert res[0].status == 2                                        # LIO_TOO_FEW_CORR: pose untouched, members untouched
    assert res[0].is_degenerate == 1
    assert np.array_equal(np.array(res[0].matP, np.float32), np.array(res_ref.matP, np.float32))
    s2m.close(); ref.close()
