"""Handle lifecycle: a SLAM node runs for hours and re-creates nothing, but a service that registers batches creates
and destroys handles all day.  Every flavour of handle must give its device memory back on destroy, survive
destroy in any state (mid-run, after an error), and leave later handles unaffected.  The reference's analogue is the
node's constructor/destructor pair (allocateMemory MO:229-262 and the pcl::PointCloud::Ptr members it resets)."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def _one_round(pkg, case, flavour):
    qs = case["queries"]
    scans = [q["scan"] for q in qs]
    poses0 = np.stack([q["pose_init"] for q in qs])
    if flavour == "plain":
        s = pkg.ScanToMap(sort_scan=2, use_graph=1, graph_iters=4)
        s.set_map(case["map"])
        s.batch_upload(scans); s.batch_set_poses(poses0); s.batch_run()
        p, _ = s.batch_results()
        s.close()
    elif flavour == "sharer":
        a = pkg.ScanToMap(sort_scan=2)
        b = pkg.ScanToMap(sort_scan=2)
        a.set_map(case["map"]); b.share_map(a)
        b.batch_upload(scans); b.batch_set_poses(poses0); b.batch_run()
        p, _ = b.batch_results()
        a.close(); b.close()                      # the owner goes first: the sharer must not touch freed memory
    elif flavour == "cert":
        s = pkg.ScanToMap(sort_scan=2, pipeline=1)
        s.set_map(case["map"])
        s.batch_upload(scans); s.batch_set_poses(poses0); s.batch_run()
        p, _ = s.batch_results()
        s.close()
    elif flavour == "multi":
        s = pkg.ScanToMap(sort_scan=2, n_devices=2, device_ids=[0, 0])
        s.set_map(case["map"])
        s.batch_upload(scans); s.batch_set_poses(poses0); s.batch_run()
        p, _ = s.batch_results()
        s.close()
    elif flavour == "abandoned":
        s = pkg.ScanToMap(sort_scan=2, use_graph=1, graph_iters=4)
        s.set_map(case["map"])
        s.batch_upload(scans); s.batch_set_poses(poses0); s.batch_run()
        s.close()                                 # destroyed with the launch loop still in flight
        p = None
    elif flavour == "store":
        st = pkg.KeyframeStore()
        rec = np.zeros((len(case["map"]), 4), np.float32)
        rec[:, :3] = case["map"]
        for k in range(4):
            st.add(rec[k::4])
        s = pkg.ScanToMap(sort_scan=2)
        ident = np.zeros((4, 6), np.float32)
        st.assemble(list(range(4)), ident, 0.4, s2m=s, want_output=False)
        p, _, _ = s.scan2MapOptimization(scans[0], poses0[0])
        s.close(); st.close()
    return p


@pytest.mark.parametrize("flavour", ["plain", "sharer", "cert", "multi", "abandoned", "store"])
def test_destroy_returns_device_memory(pkg, small_case, flavour):
    first = _one_round(pkg, small_case, flavour)   # warm-up: code objects, HIP's own pools, pinned staging
    _one_round(pkg, small_case, flavour)
    gc.collect()
    before = _free_bytes()
    for _ in range(12):
        again = _one_round(pkg, small_case, flavour)
        if first is not None:
            np.testing.assert_array_equal(again, first)
    gc.collect()
    after = _free_bytes()
    # a leak of one handle's buffers is tens of MiB per round; allow allocator granularity only
    assert before - after < 8 << 20, (flavour, before - after)


def test_error_then_reuse(pkg, small_case):
    """A refused call leaves the handle usable (the node keeps its handle across a bad message)."""
    qs = small_case["queries"]
    s = pkg.ScanToMap(sort_scan=2)
    with pytest.raises(pkg.LioError):
        s.scan2MapOptimization(qs[0]["scan"], qs[0]["pose_init"])     # LIO_ERR_NO_MAP
    s.set_map(small_case["map"])
    with pytest.raises(pkg.LioError):
        s.batch_upload([qs[0]["scan"]]); s.batch_run()                # poses not set: LIO_ERR_ARG
    p, r, _ = s.scan2MapOptimization(qs[0]["scan"], qs[0]["pose_init"])
    ref = pkg.ScanToMap(sort_scan=2)
    ref.set_map(small_case["map"])
    pw, rw, _ = ref.scan2MapOptimization(qs[0]["scan"], qs[0]["pose_init"])
    np.testing.assert_array_equal(p, pw)
    assert r.iters == rw.iters
    s.close(); ref.close()


def test_sharer_follows_an_asynchronously_installed_map(pkg, small_case):
    """lio_assemble_map_resident returns with the owner's grid build still in flight on the owner's stream; a handle
    sharing that map runs on its own stream and must wait for the build (event), not read a half-built grid."""
    qs = small_case["queries"]
    m = small_case["map"]
    rec = np.zeros((len(m), 4), np.float32)
    rec[:, :3] = m
    st = pkg.KeyframeStore()
    ids = [st.add(rec[k::5]) for k in range(5)]
    ident = np.zeros((5, 6), np.float32)
    owner = pkg.ScanToMap()
    sharer = pkg.ScanToMap()
    sharer.share_map(owner)
    ref = pkg.ScanToMap()
    for rep in range(6):
        leaf = 0.4 + 0.05 * (rep % 3)                       # a different map every time
        out, n_out, _ = st.assemble(ids, ident, leaf, s2m=None, want_output=True, max_out=len(m))
        st.assemble(ids, ident, leaf, s2m=owner, want_output=False)          # asynchronous installation
        p_sh, r_sh, _ = sharer.scan2MapOptimization(qs[rep % len(qs)]["scan"], qs[rep % len(qs)]["pose_init"])
        ref.set_map(np.ascontiguousarray(out[:n_out, :3]))
        p_ref, r_ref, _ = ref.scan2MapOptimization(qs[rep % len(qs)]["scan"], qs[rep % len(qs)]["pose_init"])
        np.testing.assert_array_equal(p_sh, p_ref)
        assert r_sh.iters == r_ref.iters and list(r_sh.n_corr_iter) == list(r_ref.n_corr_iter)
    for h in (sharer, owner, ref):
        h.close()
    st.close()
