"""In-library multi-GPU mode (cfg.n_devices > 1): one handle, the map cut into slabs over the listed
devices, the per-scan sums joined once per Gauss-Newton iteration inside the library -- what the OpenMP barrier of
MO:1622-1686 is to the reference.  The test box has one GPU, so the same ordinal is listed several times: every code
path (slab plan, halo, owner test, workgroup cull, exchange, identical solve) runs; only the peer traffic is local."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL_T, TOL_R = 1e-5, 1e-6


@pytest.mark.parametrize("n_dev", [2, 3, 5])          # (4 and more: the publish events are joined on one stream)
def test_multi_device_handle_matches_single_device(pkg, oracle, small_case, n_dev):
    one = pkg.ScanToMap(record_corr_iter=0)
    one.set_map(small_case["map"])
    many = pkg.ScanToMap(record_corr_iter=0, n_devices=n_dev, device_ids=[0] * n_dev)
    many.set_map(small_case["map"])
    for q in small_case["queries"]:
        p1, r1, rc1 = one.scan2MapOptimization(q["scan"], q["pose_init"])
        c1 = one.get_correspondences(0)
        p2, r2, rc2 = many.scan2MapOptimization(q["scan"], q["pose_init"])
        c2 = many.get_correspondences(0)
        assert rc1 == rc2 == 0 and r1.iters == r2.iters and r1.converged == r2.converged and r1.is_degenerate == r2.is_degenerate
        assert list(r1.n_corr_iter) == list(r2.n_corr_iter)
        for a, b in zip(c1, c2):                           # iteration-0 association: the union over the devices, bit for bit
            np.testing.assert_array_equal(a, b)
        assert np.abs(p1[3:] - p2[3:]).max() <= TOL_T and np.abs(p1[:3] - p2[:3]).max() <= TOL_R
        po, ro, _, _ = oracle.scan2map(oracle.default_config(knn_mode=1, n_threads=8), q["scan"], small_case["map"], q["pose_init"])
        assert ro.iters == r2.iters and np.abs(p2[3:] - po[3:]).max() <= TOL_T and np.abs(p2[:3] - po[:3]).max() <= TOL_R
    one.close(); many.close()


def test_multi_device_batches_are_reproducible_and_ragged(pkg, small_case):
    qs = small_case["queries"]
    scans = [q["scan"] for q in qs] + [qs[0]["scan"][:20], qs[1]["scan"][::3]]
    poses0 = np.stack([q["pose_init"] for q in qs] + [qs[0]["pose_init"], qs[1]["pose_init"]])
    ref = pkg.ScanToMap()
    ref.set_map(small_case["map"])
    ref.batch_upload(scans); ref.batch_set_poses(poses0); ref.batch_run()
    pr, rr = ref.batch_results()
    m = pkg.ScanToMap(n_devices=2, device_ids=[0, 0])
    m.set_map(small_case["map"])
    outs = []
    for rep in range(2):
        m.batch_upload(scans); m.batch_set_poses(poses0); m.batch_run()
        outs.append(m.batch_results())
    np.testing.assert_array_equal(outs[0][0], outs[1][0])          # sums are added in device order: run to run identical
    pm, rm = outs[0]
    assert [r.iters for r in rm] == [r.iters for r in rr] and [r.status for r in rm] == [r.status for r in rr]
    assert rm[len(qs)].status == 1                                    # too few points, MO:1844
    np.testing.assert_allclose(pm[:, 3:], pr[:, 3:], atol=TOL_T)
    np.testing.assert_allclose(pm[:, :3], pr[:, :3], atol=TOL_R)
    # the front handle refuses what only makes sense per device
    with pytest.raises(pkg.LioError):
        m.set_shard(0, 0, 1)
    # a new map re-plans the slabs
    m.set_map(small_case["map"][::2]); m.batch_set_poses(poses0); m.batch_run()
    ref.set_map(small_case["map"][::2]); ref.batch_set_poses(poses0); ref.batch_run()
    np.testing.assert_allclose(m.batch_results()[0], ref.batch_results()[0], atol=TOL_T)
    ref.close(); m.close()


def test_multi_device_with_a_smaller_gate_keeps_exact_neighbours(pkg, oracle, small_case):
    """A non-default max_sq_dist changes the cell of the slab plan and of the owner test together (they must agree, or
    points end up unowned or doubly owned)."""
    q = small_case["queries"][0]
    kw = dict(max_sq_dist=0.49, record_corr_iter=1)
    one = pkg.ScanToMap(**kw); one.set_map(small_case["map"])
    many = pkg.ScanToMap(n_devices=2, device_ids=[0, 0], **kw); many.set_map(small_case["map"])
    p1, r1, _ = one.scan2MapOptimization(q["scan"], q["pose_init"])
    p2, r2, _ = many.scan2MapOptimization(q["scan"], q["pose_init"])
    assert list(r1.n_corr_iter) == list(r2.n_corr_iter) and r1.iters == r2.iters
    for a, b in zip(one.get_correspondences(0), many.get_correspondences(0)):
        np.testing.assert_array_equal(a, b)
    one.close(); many.close()


def test_device_side_exchange_issues_no_stream_synchronisation(pkg, small_case, monkeypatch):
    """The join of the per-scan sums (MO:1622-1686) happens on the devices: peer stores into every device's gather buffer,
    cross-stream event waits, slots added in device order inside the solving kernel.  The host never synchronises a
    stream inside the Gauss-Newton loop (lio_s2m_profile.multi_stream_syncs counts them) and looks at a convergence count
    only `lookahead` iterations late; the copy form (no peer access) and the round-2 host form give the same bits."""
    qs = small_case["queries"]
    scans = [q["scan"] for q in qs]
    poses0 = np.stack([q["pose_init"] for q in qs])
    outs = {}
    for ex in ("", "copy", "host"):
        if ex:
            monkeypatch.setenv("LIO_MULTI_EXCHANGE", ex)
        else:
            monkeypatch.delenv("LIO_MULTI_EXCHANGE", raising=False)
        m = pkg.ScanToMap(n_devices=3, device_ids=[0, 0, 0])
        m.set_map(small_case["map"])
        m.batch_upload(scans); m.batch_set_poses(poses0); m.batch_run()
        p, r = m.batch_results()
        pr = m.profile()
        outs[ex] = (p, [x.iters for x in r], np.stack([np.array(x.matP) for x in r]), np.stack([np.array(x.AtA) for x in r]))
        assert pr.multi_exchange == {"": 0, "copy": 1, "host": 2}[ex]
        assert pr.multi_iterations >= max(outs[ex][1])
        if ex != "host":
            assert pr.multi_stream_syncs == 0
            # lookahead 2: the first count looked at is the one after iteration 0, when iteration 3 is about to be enqueued
            assert pr.multi_event_waits == pr.multi_iterations - 3 + (1 if pr.multi_iterations < m.cfg.max_iters else 0) or pr.multi_iterations <= 3
        else:
            assert pr.multi_stream_syncs == 3 * pr.multi_iterations
        m.close()
    for ex in ("copy", "host"):
        for a, b in zip(outs[""], outs[ex]):
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


def test_multi_device_lookahead_never_changes_a_result(pkg, small_case):
    qs = small_case["queries"]
    scans = [q["scan"] for q in qs]
    poses0 = np.stack([q["pose_init"] for q in qs])
    ref = {}
    for look, nd in ((0, 2), (1, 4), (4, 2), (2, 4)):      # (four devices: the publish events are joined on one stream)
        m = pkg.ScanToMap(n_devices=nd, device_ids=[0] * nd, lookahead=look)
        m.set_map(small_case["map"])
        for rep in range(2):                               # the second run re-uses the gather buffers and events
            m.batch_upload(scans); m.batch_set_poses(poses0); m.batch_run()
            p, r = m.batch_results()
            got = (p, [x.iters for x in r], [list(x.n_corr_iter) for x in r])
            ref.setdefault(nd, got)
            np.testing.assert_array_equal(ref[nd][0], got[0])         # same device count: the same bits whatever the lookahead
            assert ref[nd][1:] == got[1:]
        m.close()
    np.testing.assert_allclose(ref[2][0], ref[4][0], atol=TOL_T)      # another partition of the sums: within tolerance
    assert ref[2][1:] == ref[4][1:]
