#!/usr/bin/env python3
"""Single-registration latency (lio_s2m_register incl. H2D of the scan and D2H of the result) for the launch-loop options
(cfg.pipeline = 1: eager launches with a lookahead, hipGraph chunks) and the one-launch loop (cfg.pipeline = 4; what the default
configuration picks for a lone registration); the number in brackets is the pipeline the library reports,
against the 200-keyframe map of the bench case when its cache is given:  python tools/latency_sweep.py [case.npz]"""
import importlib, os, sys, time
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
print(f"# lio_s2m_register, hdl64 scans N_s ~ {np.mean([len(q['scan']) for q in qs]):.0f} vs N_m = {len(map_xyz)}, incl. H2D of the scan and D2H of the result")
for name, cfg in [("eager look=1", dict(pipeline=1, lookahead=1)), ("eager look=2", dict(pipeline=1, lookahead=2)), ("eager look=3", dict(pipeline=1, lookahead=3)),
                  ("graph 3", dict(pipeline=1, use_graph=1, graph_iters=3)), ("graph 4", dict(pipeline=1, use_graph=1, graph_iters=4)),
                  ("graph 6", dict(pipeline=1, use_graph=1, graph_iters=6)), ("graph 8", dict(pipeline=1, use_graph=1, graph_iters=8)),
                  ("one launch", dict(pipeline=4)), ("default cfg", dict())]:
    s2m = pkg.ScanToMap(**cfg)
    s2m.set_map(map_xyz)
    for q in qs[:4]:
        s2m.scan2MapOptimization(q["scan"], q["pose_init"])
    reps = 5
    iters, ts = [], []
    for _ in range(reps):
        for q in qs:
            t = time.perf_counter()
            _, res, _ = s2m.scan2MapOptimization(q["scan"], q["pose_init"])
            ts.append(time.perf_counter() - t)
            iters.append(res.iters)
    # median of the per-call times: one config of a long sweep once showed 0.6-1.2 ms per call in the MEAN while every call but
    # one or two took 0.16 ms (a host-side pause of ~50-100 ms inside the timed loop -- the interpreter's collector; it followed
    # the position in the sweep, not the configuration, and never appeared with per-call timing)
    dt = 1e3 * float(np.median(ts))
    name = name + f" [{s2m.profile().pipeline}]" + (f" fallbacks {s2m.profile().persist_fallbacks}" if s2m.profile().persist_fallbacks else "")
    print(f"{name:16s} {dt:.3f} ms per registration, median of 80 calls (mean GN iterations {np.mean(iters):.2f}, {1e3 * dt / np.mean(iters):.1f} us per iteration all included)")
    s2m.close()
