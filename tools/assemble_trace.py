#!/usr/bin/env python3
"""Local-map assembly from resident keyframes (lio_assemble_map_resident + grid build), 200 keyframes of the bench case:
wall time per call; run under `rocprofv3 --kernel-trace` and summarise with tools/prof_summary.py to split it into kernels
and host.   python tools/assemble_trace.py"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("lio-slam_amd")
synth = importlib.import_module("lio-slam_amd.synth")
case = synth.make_case("hdl64", n_keyframes=200, seed=synth.BASE_SEED, n_queries=1, device="cuda", workers=8)
kc = [np.concatenate([c, np.zeros((len(c), 1), np.float32)], 1) for c, _ in case["keyframes"]]
kp = np.stack([p for _, p in case["keyframes"]])
node = pkg.ScanToMap()
store = pkg.KeyframeStore()
ids = [store.add(c) for c in kc]
for _ in range(3):
    store.assemble(ids, kp, 0.5, s2m=node, want_output=False)
t0 = time.perf_counter()
n = 20
for _ in range(n):
    _, n_out, _ = store.assemble(ids, kp, 0.5, s2m=node, want_output=False)
node.profile()                       # (waits for the last build, which is still in flight when assemble returns)
print(f"assemble: {1e3 * (time.perf_counter() - t0) / n:.3f} ms per call incl. the grid build, {sum(len(c) for c in kc)} points in, {n_out} out")
