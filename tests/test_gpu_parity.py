"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on the same inputs.

Bar (SURVEY 8c / north_star): correspondence flags, neighbour index sets and
plane coefficients BIT-EXACT per iteration-0 association; iteration count,
convergence and degeneracy flags equal; final pose within POSE_TOL.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# The normal equations are accumulated in fp64 from exact fp32 products on both
# sides (only the summation order differs), so after rounding to fp32 they are
# normally bit-identical; the tolerance covers the rare rounding-boundary case.
POSE_TOL_T = 1e-5   # metres
POSE_TOL_R = 1e-6   # radians


def _run_both(pkg, oracle, scan, map_xyz, pose0, corr_iter=0, **cfg):
    s2m = pkg.ScanToMap(record_corr_iter=corr_iter, **cfg)
    s2m.set_map(map_xyz)
    pose, res, rc = s2m.scan2MapOptimization(scan, pose0)
    corr = s2m.get_correspondences(0)
    ocfg = oracle.default_config(knn_mode=1, n_threads=8,
                                 **{k: v for k, v in cfg.items() if k in ("jacobian_mode", "force_all_iters", "max_iters")})
    if "cell_size" in cfg:
        pass   # grid granularity never changes results (exact search)
    pose_o, res_o, matP_o, corr_o = oracle.scan2map(ocfg, scan, map_xyz, pose0, corr_iter=corr_iter)
    s2m.close()
    return (pose, res, rc, corr), (pose_o, res_o, matP_o, corr_o)


def _assert_parity(g, o):
    (pose, res, rc, corr), (pose_o, res_o, matP_o, corr_o) = g, o
    assert rc == res_o.status
    assert res.iters == res_o.iters
    assert res.converged == res_o.converged
    assert res.is_degenerate == res_o.is_degenerate
    assert list(res.n_corr_iter)[:res.iters] == list(res_o.n_corr_iter)[:res_o.iters]
    flag, coeff, nn = corr
    flag_o, coeff_o, nn_o = corr_o
    assert np.array_equal(flag, flag_o), f"{int((flag != flag_o).sum())} correspondence flags differ"
    assert np.array_equal(nn, nn_o), "5-NN index sets differ"
    sel = flag == 1
    assert np.array_equal(coeff[sel].view(np.uint32), coeff_o[sel].view(np.uint32)), "coefficients not bit-exact"
    assert np.abs(pose[3:] - pose_o[3:]).max() <= POSE_TOL_T
    assert np.abs(pose[:3] - pose_o[:3]).max() <= POSE_TOL_R
    if res.iters > 0 and rc == 0:
        # the eigen-decomposition / inverse / product chain of the first iteration (MO:1786-1808), bit for bit
        assert np.array_equal(np.array(res.matP, np.float32).view(np.uint32), np.asarray(matP_o, np.float32).reshape(-1).view(np.uint32)), \
            "matP differs from the oracle"


def test_register_matches_oracle(pkg, oracle, small_case):
    for q in small_case["queries"]:
        g, o = _run_both(pkg, oracle, q["scan"], small_case["map"], q["pose_init"])
        _assert_parity(g, o)
        # known answer: the true pose is recovered
        assert np.abs(g[0][3:] - q["pose_true"][3:]).max() < 0.05
        assert np.abs(g[0][:3] - q["pose_true"][:3]).max() < 0.01


@pytest.mark.parametrize("variant", [
    dict(use_lds=0, sort_scan=0), dict(use_lds=1, sort_scan=0), dict(use_lds=0, sort_scan=2),
    dict(use_lds=1, sort_scan=2, kernel_variant=2), dict(use_lds=1, sort_scan=2, kernel_variant=4),
    dict(use_lds=0, sort_scan=2, kernel_variant=4), dict(use_lds=1, sort_scan=2, cell_size=2.5),
    dict(cell_div=1), dict(cell_div=2, cell_size=1.7), dict(cell_div=3, sort_scan=0),
    dict(x_sub=1), dict(x_sub=2), dict(x_sub=4), dict(x_sub=8, cell_div=3), dict(x_sub=4, cell_div=1, cell_size=2.5),
    dict(x_sub=4, use_lds=1, sort_scan=2), dict(x_sub=4, pipeline=4), dict(max_batch=64),
    dict(tight_rows=1), dict(tight_rows=1, x_sub=4, pipeline=4), dict(tight_rows=1, x_sub=2, cell_div=3, cell_size=2.5),
    dict(tight_rows=1, pipeline=1, sort_scan=0), dict(tight_rows=-1, max_batch=64),
    dict(tight_rows=2, x_sub=4), dict(tight_rows=3, x_sub=4, pipeline=4), dict(tight_rows=3, x_sub=1, cell_div=1),
])
def test_kernel_variants_are_equivalent(pkg, oracle, small_case, variant):
    """LDS-staged vs global candidate scan, sorted vs caller-order scans, points
    per thread, coarser grid, finer x buckets of the rows (x_sub; max_batch >= 8 picks 4), the tight second row table: all exact searches -> identical
    correspondences."""
    q = small_case["queries"][2]
    g, o = _run_both(pkg, oracle, q["scan"], small_case["map"], q["pose_init"], corr_iter=1, **variant)
    _assert_parity(g, o)


def test_later_iteration_association(pkg, oracle, small_case):
    q = small_case["queries"][0]
    g, o = _run_both(pkg, oracle, q["scan"], small_case["map"], q["pose_init"], corr_iter=2)
    _assert_parity(g, o)


def test_force_all_iters_and_exact_jacobian(pkg, oracle, small_case):
    q = small_case["queries"][1]
    g, o = _run_both(pkg, oracle, q["scan"], small_case["map"], q["pose_init"], force_all_iters=1, max_iters=12)
    _assert_parity(g, o)
    assert g[1].iters == 12
    g, o = _run_both(pkg, oracle, q["scan"], small_case["map"], q["pose_init"], jacobian_mode=1)
    _assert_parity(g, o)


def test_normal_equations_bit_exact(pkg, oracle, small_case):
    q = small_case["queries"][0]
    g, o = _run_both(pkg, oracle, q["scan"], small_case["map"], q["pose_init"], max_iters=1)
    a = np.array(g[1].AtA, np.float32)
    b = np.array(o[1].AtA, np.float32)
    # fp64 accumulation of exact products: equal after rounding except at a rounding boundary
    assert (a.view(np.uint32) != b.view(np.uint32)).sum() <= 2
    np.testing.assert_allclose(a, b, rtol=2e-7)
    np.testing.assert_allclose(np.array(g[1].AtB), np.array(o[1].AtB), rtol=2e-7, atol=1e-6)


def test_too_few_points_and_too_few_corr(pkg, oracle, small_case):
    q = small_case["queries"][0]
    s2m = pkg.ScanToMap()
    s2m.set_map(small_case["map"])
    pose, res, rc = s2m.scan2MapOptimization(q["scan"][:30], q["pose_init"])   # N_s <= 30, MO:1844
    assert rc == 1 and res.iters == 0
    assert np.array_equal(pose, q["pose_init"])
    # far away from the map: no correspondences, pose untouched, 30 iterations reported (MO:1721-1724)
    far = q["pose_init"].copy()
    far[3] += 500.0
    pose, res, rc = s2m.scan2MapOptimization(q["scan"], far)
    pose_o, res_o, _, _ = oracle.scan2map(oracle.default_config(), q["scan"], small_case["map"], far)
    assert rc == 2 == res_o.status
    assert res.iters == res_o.iters == 30
    assert np.array_equal(pose, far) and np.array_equal(pose_o, far)
    s2m.close()


@pytest.mark.parametrize("cfg", [dict(), dict(max_batch=64), dict(tight_rows=3, x_sub=8)])
def test_empty_and_tiny_maps(pkg, oracle, small_case, cfg):
    q = small_case["queries"][0]
    s2m = pkg.ScanToMap(**cfg)
    s2m.set_map(np.zeros((0, 3), np.float32))
    pose, res, rc = s2m.scan2MapOptimization(q["scan"], q["pose_init"])
    assert rc == 2 and np.array_equal(pose, q["pose_init"])
    s2m.set_map(small_case["map"][:3])            # fewer than 5 map points
    pose, res, rc = s2m.scan2MapOptimization(q["scan"], q["pose_init"])
    assert rc == 2 and np.array_equal(pose, q["pose_init"])
    bad = np.full((40, 3), np.nan, np.float32)    # nothing but non-finite points: an empty grid
    s2m.set_map(bad)
    pose, res, rc = s2m.scan2MapOptimization(q["scan"], q["pose_init"])
    assert rc == 2 and np.array_equal(pose, q["pose_init"])
    s2m.set_map(small_case["map"])                # and the handle is as good as new afterwards
    pose, res, rc = s2m.scan2MapOptimization(q["scan"], q["pose_init"])
    pose_o, res_o, _, _ = oracle.scan2map(oracle.default_config(knn_mode=1, n_threads=8), q["scan"], small_case["map"], q["pose_init"])
    assert rc == 0 and res.iters == res_o.iters and np.abs(pose - pose_o).max() <= 1e-5
    s2m.close()


def test_batch_matches_single(pkg, oracle, small_case):
    qs = small_case["queries"]
    scans = [q["scan"] for q in qs] + [qs[0]["scan"][:20]]   # ragged batch incl. a too-small scan
    poses0 = np.stack([q["pose_init"] for q in qs] + [qs[0]["pose_init"]])
    s2m = pkg.ScanToMap(max_batch=64)
    s2m.set_map(small_case["map"])
    s2m.batch_upload(scans)
    s2m.batch_set_poses(poses0)
    s2m.batch_run()
    poses, res = s2m.batch_results()
    for i, q in enumerate(qs):
        pose_o, res_o, _, _ = oracle.scan2map(oracle.default_config(), q["scan"], small_case["map"], q["pose_init"])
        assert res[i].iters == res_o.iters and res[i].status == res_o.status
        assert np.abs(poses[i][3:] - pose_o[3:]).max() <= POSE_TOL_T
        assert np.abs(poses[i][:3] - pose_o[:3]).max() <= POSE_TOL_R
    assert res[len(qs)].status == 1 and np.array_equal(poses[len(qs)], poses0[len(qs)])
    # re-running the same batch is bit-reproducible (fixed reduction trees)
    s2m.batch_set_poses(poses0)
    s2m.batch_run()
    poses2, _ = s2m.batch_results()
    assert np.array_equal(poses, poses2)
    s2m.close()


def test_pcl_stride_input(pkg, small_case):
    """pcl::PointXYZI layout (32-byte stride) gives the same result as packed xyz."""
    q = small_case["queries"][0]
    def pcl(a):
        out = np.zeros((len(a), 8), np.float32)
        out[:, :3] = a
        out[:, 3] = 1.0
        out[:, 4] = 42.0
        return out
    s2m = pkg.ScanToMap()
    s2m.set_map(small_case["map"])
    p1, r1, _ = s2m.scan2MapOptimization(q["scan"], q["pose_init"])
    s2m.set_map(pcl(small_case["map"]))
    p2, r2, _ = s2m.scan2MapOptimization(pcl(q["scan"]), q["pose_init"])
    assert np.array_equal(p1, p2) and r1.iters == r2.iters
    s2m.close()


def test_exact_ties_and_coincident_points(pkg, oracle):
    """Lattice map (many exactly equal distances) + scan points ON map points
    (d2 == 0 -> denormal fp64 keys): the (d2, index) tie-break must match."""
    g = np.arange(0, 6, 0.5, dtype=np.float32)
    lattice = np.stack(np.meshgrid(g, g, np.array([0.0, 0.5], np.float32), indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(0)
    map_xyz = np.concatenate([lattice, lattice[::5]]).astype(np.float32)     # duplicates: ties at d2 == 0 too
    scan = np.concatenate([lattice[::3] + np.float32(0.25), lattice[1::7],
                           rng.uniform(0.5, 5.5, (200, 3)).astype(np.float32) * np.array([1, 1, 0.1], np.float32)])
    pose0 = np.zeros(6, np.float32)
    for variant in (dict(use_lds=0), dict(use_lds=1)):
        g_, o_ = _run_both(pkg, oracle, scan, map_xyz, pose0, corr_iter=0, max_iters=1, **variant)
        flag, coeff, nn = g_[3]
        flag_o, coeff_o, nn_o = o_[3]
        assert np.array_equal(nn, nn_o)
        assert np.array_equal(flag, flag_o)
        assert (nn_o[:, 4] >= 0).sum() > 100


@pytest.mark.parametrize("graph_iters", [1, 4, 7, 30])
def test_hipgraph_loop_is_identical(pkg, small_case, graph_iters):
    """BASELINE config 5: the GN loop replayed from a captured hipGraph chunk gives the same bits
    as the eager launch loop, for single scans and for ragged batches, also across re-uploads."""
    qs = small_case["queries"]
    scans = [q["scan"] for q in qs] + [qs[0]["scan"][:20], qs[1]["scan"][::2]]
    poses0 = np.stack([q["pose_init"] for q in qs] + [qs[0]["pose_init"], qs[1]["pose_init"]])
    eager = pkg.ScanToMap()
    eager.set_map(small_case["map"])
    eager.batch_upload(scans); eager.batch_set_poses(poses0); eager.batch_run()
    pe, re_ = eager.batch_results()
    g = pkg.ScanToMap(use_graph=1, graph_iters=graph_iters)
    g.set_map(small_case["map"])
    for rep in range(2):                                   # second round replays the cached graph
        g.batch_upload(scans); g.batch_set_poses(poses0); g.batch_run()
        pg, rg = g.batch_results()
        np.testing.assert_array_equal(pg, pe)
        assert [r.iters for r in rg] == [r.iters for r in re_]
        assert [r.status for r in rg] == [r.status for r in re_]
    p1, r1, _ = g.scan2MapOptimization(qs[2]["scan"], qs[2]["pose_init"])      # new geometry -> re-capture
    np.testing.assert_array_equal(p1, pe[2])
    eager.close(); g.close()


def test_degeneracy_chain_is_bit_exact_on_many_normal_matrices(pkg, oracle, synth, small_case):
    """cv::eigen (Jacobi), the degeneracy rule, matV.inv() and the product (MO:1786-1808) run once per registration, inside
    the first Gauss-Newton launch, as register-resident wave code; isDegenerate and all 36 entries of matP must match the CPU
    restatement bit for bit on well-conditioned scenes, on a degenerate corridor, and with perturbed and truncated inputs
    (different normal matrices, different pivot sequences)."""
    cases = [(small_case["map"], q["scan"], q["pose_init"]) for q in small_case["queries"]]
    cor = synth.make_case("vlp16", n_keyframes=5, seed=3, kind="corridor", device="cpu")
    cases += [(cor["map"], q["scan"], q["pose_init"]) for q in cor["queries"]]
    rng = np.random.default_rng(17)
    more = []
    for m, sc, p0 in cases:
        for k in range(4):
            sub = sc[rng.permutation(len(sc))[: len(sc) // (k + 1)]]
            pp = np.array(p0, np.float32) + rng.normal(0, [0.004, 0.004, 0.01, 0.05, 0.05, 0.02]).astype(np.float32)
            more.append((m, np.ascontiguousarray(sub), pp))
    n_deg = 0
    for m, sc, p0 in cases + more:
        s2m = pkg.ScanToMap(max_iters=2)
        s2m.set_map(m)
        _, res, rc = s2m.scan2MapOptimization(sc, p0)
        s2m.close()
        _, res_o, matP_o, _ = oracle.scan2map(oracle.default_config(knn_mode=1, n_threads=8, max_iters=2), sc, m, p0)
        assert rc == res_o.status and res.is_degenerate == res_o.is_degenerate
        n_deg += res.is_degenerate
        if rc == 0:
            assert np.array_equal(np.array(res.matP, np.float32).view(np.uint32), np.asarray(matP_o, np.float32).reshape(-1).view(np.uint32))
            assert np.array_equal(np.array(res.AtA, np.float32).view(np.uint32), np.array(res_o.AtA, np.float32).view(np.uint32))
    assert n_deg >= 2


@pytest.mark.skipif(any(os.environ.get(k) for k in ("LIO_TIGHT", "LIO_X_SUB", "LIO_TRY")), reason="the A/B overrides replace the automatic choice")
def test_row_tables_follow_the_handle_and_the_map_density(pkg, oracle, small_case):
    """cfg.x_sub / cfg.tight_rows = auto: a node's handle (max_batch < 8) keeps the plain rows, a batch handle gets x buckets four
    times finer and one tight table, plus the finer tables when the map is dense enough to have queries for them -- and the
    results do not depend on any of it."""
    q = small_case["queries"][0]
    got = []
    for cfg in (dict(), dict(max_batch=64), dict(max_batch=64, tight_rows=-1, x_sub=1)):
        s2m = pkg.ScanToMap(**cfg)
        s2m.set_map(small_case["map"])
        pose, res, rc = s2m.scan2MapOptimization(q["scan"], q["pose_init"])
        prof = s2m.profile()
        got.append((pose.copy(), res.iters, prof.map_x_sub, prof.map_tight_tables, prof.map_pts_per_cell))
        s2m.close()
    assert got[0][2:4] == (1, 0) and got[0][4] == 0.0
    assert got[1][2] == 4 and got[1][3] >= 1 and got[1][4] > 1.0
    assert got[2][2:4] == (1, 0)
    for g in got[1:]:
        assert np.array_equal(g[0].view(np.uint32), got[0][0].view(np.uint32)) and g[1] == got[0][1]
    # a dense map (2 cm spacing on a few planes): all three tables
    rng = np.random.default_rng(5)
    u = np.stack(np.meshgrid(np.arange(0, 6, 0.03), np.arange(0, 6, 0.03)), -1).reshape(-1, 2).astype(np.float32)
    planes = [np.concatenate([u, np.full((len(u), 1), z, np.float32)], 1) for z in (0.0, 2.5)]
    planes.append(np.concatenate([u[:, :1], np.full((len(u), 1), 6.0, np.float32), u[:, 1:] * 0.4], 1))
    dense = (np.concatenate(planes) + rng.normal(0, 0.004, (3 * len(u), 3))).astype(np.float32)
    # ... and a query without a bound tries the finest table first, then the coarser ones, then the full search: scan points on
    # the planes (found at once), 0.1-0.5 m off them (found one or two tables up) and metres away (every table fails, then the gate)
    scan = np.concatenate([dense[rng.choice(len(dense), 4000, replace=False)] + rng.normal(0, 0.01, (4000, 3)),
                           dense[rng.choice(len(dense), 600, replace=False)] + rng.uniform(0.1, 0.5, (600, 1)) * np.array([0.0, 0.0, 1.0]),
                           rng.uniform(-4, 10, (200, 3))]).astype(np.float32)
    pose0 = np.array([0.004, -0.003, 0.005, 0.03, -0.02, 0.025], np.float32)
    for corr_iter in (0, 2):
        ref = None
        for cfg in (dict(max_batch=64), dict(tight_rows=-1, x_sub=1), dict(max_batch=64, pipeline=4)):
            s2m = pkg.ScanToMap(record_corr_iter=corr_iter, **cfg)
            s2m.set_map(dense)
            pose, res, rc = s2m.scan2MapOptimization(scan, pose0)
            corr = s2m.get_correspondences(0)
            prof = s2m.profile()
            s2m.close()
            if ref is None:
                assert prof.map_tight_tables == 3 and prof.map_first_try == 2 and prof.map_pts_per_cell > 50
                ref = (pose, res.iters, corr)
                continue
            assert prof.map_tight_tables == (0 if cfg.get("tight_rows") == -1 else 3)
            assert np.array_equal(pose.view(np.uint32), ref[0].view(np.uint32)) and res.iters == ref[1]
            for a, b in zip(corr, ref[2]):
                assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
        assert ref[1] >= 3
        # the oracle's exact search agrees (flags and neighbour sets of the recorded iteration)
        ocfg = oracle.default_config(knn_mode=1, n_threads=8)
        _, res_o, _, corr_o = oracle.scan2map(ocfg, scan, dense, pose0, corr_iter=corr_iter)
        assert res_o.iters == ref[1]
        assert np.array_equal(ref[2][0], corr_o[0]) and np.array_equal(ref[2][2], corr_o[2])
