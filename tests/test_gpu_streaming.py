"""Streaming front end of the batched path (the metric includes the per-scan H2D; BASELINE configs[4]
"streamed"): two handles sharing one resident map as a double buffer, uploads from pinned memory with the host
never touching the points, device-resident inputs, and the deterministic order of oversized sort tiles.
The reference analogue is one cloud arriving per callback, MO:432-476."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _records(scans, stride):
    """Back-to-back records of `stride` bytes (x,y,z at 0,4,8; PCL's PointXYZI is stride 32)."""
    n = sum(len(s) for s in scans)
    buf = np.zeros((n, stride // 4), np.float32)
    buf[:, :3] = np.concatenate(scans)
    if stride >= 20:
        buf[:, 4] = 7.0
    return buf


def _reference(pkg, case, scans, poses0, **cfg):
    ref = pkg.ScanToMap(**cfg)
    ref.set_map(case["map"])
    ref.batch_upload(scans); ref.batch_set_poses(poses0); ref.batch_run()
    p, r = ref.batch_results()
    ref.close()
    return p, r


@pytest.mark.parametrize("stride", [12, 16, 32])
def test_double_buffered_stream_matches_plain_batches(pkg, small_case, stride):
    qs = small_case["queries"]
    rng = np.random.default_rng(3)
    batches = []
    for b in range(5):                                  # five different ragged batches
        order = rng.permutation(len(qs))
        scans = [qs[i]["scan"][: len(qs[i]["scan"]) - 17 * b] for i in order] + [qs[0]["scan"][:25 + b]]
        poses = np.stack([qs[i]["pose_init"] for i in order] + [qs[0]["pose_init"]]).astype(np.float32)
        poses[:, 3:] += rng.normal(0, 0.02, (len(poses), 3)).astype(np.float32)
        batches.append((scans, poses))
    cfg = dict(sort_scan=2, use_graph=1, graph_iters=4)
    want = [_reference(pkg, small_case, s, p, **cfg) for s, p in batches]

    a = pkg.ScanToMap(**cfg)
    b = pkg.ScanToMap(**cfg)
    a.set_map(small_case["map"])
    b.share_map(a)
    handles = [a, b]
    pins = []
    for scans, _ in batches:
        rec = _records(scans, stride)
        pin = pkg.PinnedBuffer(rec.nbytes)
        pin.array[:] = rec.view(np.uint8).reshape(-1)
        pins.append(pin)
    # software pipeline: upload k+1 while k iterates
    got = [None] * len(batches)
    handles[0].batch_upload_raw(pins[0].ptr, [len(s) for s in batches[0][0]], stride)
    for k in range(len(batches)):
        h = handles[k % 2]
        h.batch_set_poses(batches[k][1])
        h.batch_run()                                   # asynchronous
        if k + 1 < len(batches):
            handles[(k + 1) % 2].batch_upload_raw(pins[k + 1].ptr, [len(s) for s in batches[k + 1][0]], stride)
        got[k] = h.batch_results()
    for (p, r), (pw, rw) in zip(got, want):
        np.testing.assert_array_equal(p, pw)
        assert [x.iters for x in r] == [x.iters for x in rw] and [x.status for x in r] == [x.status for x in rw]
        for x, y in zip(r, rw):
            np.testing.assert_array_equal(np.array(x.AtA, np.float32).view(np.uint32), np.array(y.AtA, np.float32).view(np.uint32))
    # a sharer cannot replace the map, and follows a new map of the owner
    with pytest.raises(pkg.LioError):
        b.set_map(small_case["map"])
    a.batch_sync(); b.batch_sync()
    half = small_case["map"][::2]
    a.set_map(half)
    b.batch_set_poses(batches[-1][1] if (len(batches) - 1) % 2 == 1 else batches[-2][1]); b.batch_run()
    pb, _ = b.batch_results()
    k_b = len(batches) - 1 if (len(batches) - 1) % 2 == 1 else len(batches) - 2
    ref = pkg.ScanToMap(**cfg)
    ref.set_map(half); ref.batch_upload(batches[k_b][0]); ref.batch_set_poses(batches[k_b][1]); ref.batch_run()
    np.testing.assert_array_equal(pb, ref.batch_results()[0])
    ref.close(); b.close(); a.close()
    for pin in pins:
        pin.close()


def test_device_resident_inputs(pkg, small_case):
    """scans[] may be device pointers: the library never reads the caller's memory on the host."""
    import torch
    qs = small_case["queries"]
    scans = [q["scan"] for q in qs]
    poses0 = np.stack([q["pose_init"] for q in qs])
    want, _ = _reference(pkg, small_case, scans, poses0, sort_scan=2)
    rec = _records(scans, 32)
    dev = torch.from_numpy(rec).cuda()
    torch.cuda.synchronize()
    s = pkg.ScanToMap(sort_scan=2)
    s.set_map(small_case["map"])
    s.batch_upload_raw(dev.data_ptr(), [len(x) for x in scans], 32, asynchronous=False)
    s.batch_set_poses(poses0); s.batch_run()
    got, _ = s.batch_results()
    np.testing.assert_array_equal(got, want)
    s.close()


def test_oversized_tiles_are_ordered_deterministically(pkg, oracle, synth):
    """More than 1024 points in one sort tile (raw sweeps, tiny leaf sizes): the tile's points are ordered by caller
    index by k_scan_tile_bigsort, not by the arrival order of atomics.  With ONE tile per scan the sorted order is
    the caller's order, so the tile-sorted run must reproduce the unsorted run bit for bit -- and itself."""
    case = synth.make_case("vlp16", n_keyframes=5, seed=4, device="cpu", n_queries=2)
    boxes = case["boxes"]
    raws = []
    for q in case["queries"]:                            # raw (not voxel-filtered) sweeps: ~25 k points each
        sc = synth.cast_scan(boxes, q["pose_true"], "vlp16", seed=99, device="cpu")
        raws.append(np.ascontiguousarray(sc["xyz"]))
    poses0 = np.stack([q["pose_init"] for q in case["queries"]])
    outs = []
    for cfg in (dict(sort_scan=0), dict(sort_scan=2, tile_size=1.0e4), dict(sort_scan=2, tile_size=1.0e4),
                dict(sort_scan=2, tile_size=16.0), dict(sort_scan=2, tile_size=16.0)):
        s = pkg.ScanToMap(**cfg)
        s.set_map(case["map"])
        s.batch_upload(raws); s.batch_set_poses(poses0); s.batch_run()
        p, r = s.batch_results()
        outs.append((p, [np.array(x.AtA, np.float32).view(np.uint32) for x in r], [x.iters for x in r]))
        s.close()
    assert min(len(x) for x in raws) > 4096
    for k in (1, 2):                                     # one tile per scan == caller order
        np.testing.assert_array_equal(outs[k][0], outs[0][0])
        for a, b in zip(outs[k][1], outs[0][1]):
            np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(outs[3][0], outs[4][0])    # 16 m tiles (thousands of points each): run to run
    for a, b in zip(outs[3][1], outs[4][1]):
        np.testing.assert_array_equal(a, b)
    assert outs[3][2] == outs[0][2]
    np.testing.assert_allclose(outs[3][0], outs[0][0], atol=1e-5)
    po, ro, _, _ = oracle.scan2map(oracle.default_config(knn_mode=1, n_threads=8), raws[0], case["map"], poses0[0])
    assert np.abs(outs[0][0][0][3:] - po[3:]).max() <= 1e-5 and np.abs(outs[0][0][0][:3] - po[:3]).max() <= 1e-6


def test_one_launch_lds_sort_equals_the_counting_sort(pkg, small_case):
    """Scans of at most 16384 points are reordered by ONE launch (k_scan_sort_lds: box, tile keys, bitonic sort in LDS,
    gather); it must produce the permutation of the multi-kernel counting sort + rank sort (cfg.sort_scan = 3), so every
    sum -- and with it every bit of the result -- is the same.  Ragged batch, tiny and empty-ish scans included."""
    qs = small_case["queries"]
    scans = [q["scan"] for q in qs] + [qs[0]["scan"][:20], qs[1]["scan"][::2], qs[2]["scan"][:1], qs[0]["scan"][:257]]
    poses0 = np.stack([q["pose_init"] for q in qs] + [qs[0]["pose_init"], qs[1]["pose_init"], qs[2]["pose_init"], qs[0]["pose_init"]])
    outs = []
    for mode in (2, 3):
        for tile in (0.0, 1.5):
            s = pkg.ScanToMap(sort_scan=mode, tile_size=tile, record_corr_iter=1)
            s.set_map(small_case["map"])
            s.batch_upload(scans); s.batch_set_poses(poses0); s.batch_run()
            p, r = s.batch_results()
            outs.append((mode, tile, p, [np.array(x.AtA, np.float32).view(np.uint32) for x in r], s.get_correspondences(1)))
            s.close()
    for tile in (0.0, 1.5):
        a = next(o for o in outs if o[0] == 2 and o[1] == tile)
        b = next(o for o in outs if o[0] == 3 and o[1] == tile)
        np.testing.assert_array_equal(a[2], b[2])
        for u, v in zip(a[3], b[3]):
            np.testing.assert_array_equal(u, v)
        for u, v in zip(a[4], b[4]):
            np.testing.assert_array_equal(u, v)


def test_lds_sort_size_boundaries(pkg, small_case):
    """The one-launch sort (one workgroup per scan: radix sort with the keys in registers, 16 rows of 64 per wave up to 8192
    points, 32 rows above) takes scans of up to 16383 points; one point more and the batch goes through the multi-kernel
    counting sort.  Same results on both sides of every boundary."""
    rng = np.random.default_rng(9)
    m = small_case["map"]
    base = small_case["queries"][0]
    pose0 = base["pose_init"]
    big = m[rng.integers(0, len(m), 20000)] + rng.normal(0, 0.03, (20000, 3)).astype(np.float32)    # map-like points, world frame
    T = np.eye(4)
    for n in (1, 63, 64, 65, 4096, 4097, 8192, 8193, 16383, 16384, 16385):
        scan = np.ascontiguousarray(big[:n], np.float32)
        zero = np.zeros(6, np.float32)
        outs = []
        for mode in (2, 3):
            s = pkg.ScanToMap(sort_scan=mode, max_iters=3, force_all_iters=1)
            s.set_map(m)
            s.batch_upload([scan, base["scan"]]); s.batch_set_poses(np.stack([zero, pose0])); s.batch_run()
            p, r = s.batch_results()
            outs.append((p, [np.array(x.AtA, np.float32).view(np.uint32) for x in r], [x.n_corr_last for x in r]))
            s.close()
        np.testing.assert_array_equal(outs[0][0], outs[1][0])
        for a, b in zip(outs[0][1], outs[1][1]):
            np.testing.assert_array_equal(a, b)
        assert outs[0][2] == outs[1][2], n
