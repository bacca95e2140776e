#!/usr/bin/env python3
"""Host wall time of every call of one batch step on ONE handle (records resident on the device, 512 scans of the bench's case
cache): what the host spends before the GPU can start a batch, i.e. what a pipeline of several handles has to hide.
   python tools/host_cost.py <case.npz>     (case cache of `python bench.py --case-cache`)"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("lio-slam_amd")
z = np.load(sys.argv[1])
lens = z["lens"][:512]; offs = np.concatenate([[0], np.cumsum(lens)]); cat = z["scans"]
recs = np.zeros((int(offs[-1]), 8), np.float32); recs[:, :3] = cat[:offs[-1]]; recs[:, 3] = 1.0
dev = pkg.DeviceBuffer(recs)
h = pkg.ScanToMap(max_batch=512, use_graph=1, graph_iters=12, profile=0)
h.set_map(z["map"])
poses = z["poses0"][:512]
T = {k: [] for k in ("upload", "poses", "run", "results")}
for it in range(12):
    t0 = time.perf_counter(); h.batch_upload_raw(dev.ptr, list(map(int, lens)), 32)
    t1 = time.perf_counter(); h.batch_set_poses(poses)
    t2 = time.perf_counter(); h.batch_run()
    t3 = time.perf_counter(); h.batch_results(with_results=False)
    t4 = time.perf_counter()
    if it >= 2:
        T["upload"].append(t1 - t0); T["poses"].append(t2 - t1); T["run"].append(t3 - t2); T["results"].append(t4 - t3)
print({k: round(1e3 * float(np.median(v)), 4) for k, v in T.items()}, "ms per call (host wall), 512 scans")
