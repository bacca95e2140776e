#!/usr/bin/env python3
"""What a one-launch loop that cannot make progress costs: a workgroup is withheld (test hook), its scan's other workgroups
give up after `spin_max` polls, the call re-runs the registration through the launch loop and returns the same result.
   python tools/persist_timeout.py"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("lio-slam_amd")
synth = importlib.import_module("lio-slam_amd.synth")
case = synth.make_case("hdl64", n_keyframes=60, n_queries=4)
qs, map_xyz = case["queries"], case["map"]
h = pkg.ScanToMap(pipeline=4)
h.set_map(map_xyz)
for q in qs:
    h.scan2MapOptimization(q["scan"], q["pose_init"])
for spin in (0, 1024, 256):
    h.debug_persist_spin(spin_max=spin, withhold_wg=3)
    ts = []
    for q in qs:
        t = time.perf_counter(); h.scan2MapOptimization(q["scan"], q["pose_init"]); ts.append(1e3 * (time.perf_counter() - t))
    print("spin_max", spin or "default(4096)", "call with a withheld workgroup: %.2f ms (incl. the launch-loop re-run)" % np.median(ts), "fallbacks", h.profile().persist_fallbacks)
h.debug_persist_spin(0, -1)
t = time.perf_counter(); h.scan2MapOptimization(qs[0]["scan"], qs[0]["pose_init"]); print("healthy again: %.3f ms" % (1e3 * (time.perf_counter() - t)))
