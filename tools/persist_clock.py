#!/usr/bin/env python3
"""Phase clock of the one-launch Gauss-Newton loop (k_s2m_persist, cfg.pipeline = 4, cfg.profile = 2): where a lone
registration's time goes, per workgroup (wave 0), summed over the iterations.  python tools/persist_clock.py [case.npz]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("lio-slam_amd")
synth = importlib.import_module("lio-slam_amd.synth")
if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
    z = np.load(sys.argv[1])
    lens = z["lens"][:16]
    offs = np.concatenate([[0], np.cumsum(lens)])
    cat = z["scans"]
    qs = [{"scan": np.ascontiguousarray(cat[offs[i]:offs[i + 1]]), "pose_init": z["poses0"][i]} for i in range(16)]
    map_xyz = z["map"]
else:
    case = synth.make_case("hdl64", n_keyframes=60, n_queries=16)
    qs, map_xyz = case["queries"], case["map"]
names = ["state+transform", "candidate scan", "plane+row", "sums", "partial+arrival", "solve (solver only)", "barrier wait", "acquire"]
s2m = pkg.ScanToMap(pipeline=4, profile=2)
s2m.set_map(map_xyz)
rows, iters, sol = [], [], []
for rep in range(3):
    for q in qs:
        _, res, _ = s2m.scan2MapOptimization(q["scan"], q["pose_init"])
        raw = s2m.debug_stamps().astype(np.float64)
        st = raw[:, 0, :] * 0.01                                            # us per workgroup and phase
        if rep:
            rows.append(st.mean(axis=0)); iters.append(res.iters)
            sol.append(raw[:, 1, :].sum(axis=0))                            # solver detail, summed over the workgroups (ticks; [6] = count)
rows = np.array(rows)
print(f"# k_s2m_persist phase clock, {len(rows)} registrations, mean GN iterations {np.mean(iters):.2f}; us per registration, mean over the workgroups (wave 0)")
for k, n in enumerate(names):
    print(f"{n:22s} {rows[:, k].mean():7.1f} us   ({rows[:, k].mean() / np.mean(iters):5.2f} us per iteration)")
print(f"{'sum':22s} {rows.sum(axis=1).mean():7.1f} us")
sol = np.array(sol)
n_later = sol[:, 6].mean()
print(f"# the solving workgroup, per solve: iteration 0: gather of the partial sums {sol[:, 0].mean() * 0.01:.1f} us, lio_gn_step {sol[:, 1].mean() * 0.01:.1f} us, release {sol[:, 2].mean() * 0.01:.1f} us;"
      f" later iterations ({n_later:.2f} per registration): gather {sol[:, 3].mean() * 0.01 / n_later:.1f} us, lio_gn_step {sol[:, 4].mean() * 0.01 / n_later:.1f} us, release {sol[:, 5].mean() * 0.01 / n_later:.1f} us")
s2m.close()
