#!/usr/bin/env python3
"""Single-registration latency (lio_s2m_register incl. H2D of the scan and D2H of the result) for the launch-loop options."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("lio-slam_amd")
synth = importlib.import_module("lio-slam_amd.synth")
case = synth.make_case("hdl64", n_keyframes=60, n_queries=16)
qs = case["queries"]
for name, cfg in [("eager look=1", dict(lookahead=1)), ("eager look=2", dict(lookahead=2)), ("eager look=3", dict(lookahead=3)),
                  ("graph 3", dict(use_graph=1, graph_iters=3)), ("graph 4", dict(use_graph=1, graph_iters=4)),
                  ("graph 6", dict(use_graph=1, graph_iters=6)), ("graph 8", dict(use_graph=1, graph_iters=8))]:
    s2m = pkg.ScanToMap(**cfg)
    s2m.set_map(case["map"])
    for q in qs[:4]:
        s2m.scan2MapOptimization(q["scan"], q["pose_init"])
    t = time.perf_counter()
    reps = 5
    iters = []
    for _ in range(reps):
        for q in qs:
            _, res, _ = s2m.scan2MapOptimization(q["scan"], q["pose_init"])
            iters.append(res.iters)
    print(f"{name:14s} {1e3 * (time.perf_counter() - t) / (reps * len(qs)):.3f} ms per registration (mean GN iterations {np.mean(iters):.2f})")
    s2m.close()
