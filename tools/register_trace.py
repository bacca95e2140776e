#!/usr/bin/env python3
"""Timeline of ONE lio_s2m_register call (upload, tile sort, poses, init, Gauss-Newton loop, results) from a rocprofv3
--kernel-trace run:
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/register_trace.py --run PIPELINE [case.npz]
    python3 tools/register_trace.py --parse DIR
--run registers 16 scans 3 times with cfg.pipeline = PIPELINE (1 = launch loop, 4 = one launch); --parse groups the
kernels into registrations (a gap of more than 60 us of GPU idle time separates two) and prints the median timeline."""
import csv, glob, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "--run":
    pkg = importlib.import_module("lio-slam_amd")
    synth = importlib.import_module("lio-slam_amd.synth")
    pipe = int(sys.argv[2])
    if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):
        z = np.load(sys.argv[3])
        lens = z["lens"][:16]
        offs = np.concatenate([[0], np.cumsum(lens)])
        cat = z["scans"]
        qs = [{"scan": np.ascontiguousarray(cat[offs[i]:offs[i + 1]]), "pose_init": z["poses0"][i]} for i in range(16)]
        map_xyz = z["map"]
    else:
        case = synth.make_case("hdl64", n_keyframes=60, n_queries=16)
        qs, map_xyz = case["queries"], case["map"]
    s2m = pkg.ScanToMap(pipeline=pipe)
    s2m.set_map(map_xyz)
    import time
    for rep in range(3):
        for q in qs:
            s2m.scan2MapOptimization(q["scan"], q["pose_init"])
            time.sleep(0.001)                          # an idle gap between two registrations (for --parse)
    s2m.close()
    sys.exit(0)

rows = []
for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
groups, cur = [], []
for r in rows:
    if cur and r[0] - cur[-1][1] > 60000:
        groups.append(cur); cur = []
    cur.append(r)
if cur:
    groups.append(cur)
groups = [g for g in groups if any("k_s2m_" in k[2] for k in g) and not any("k_map_" in k[2] for k in g)]
groups = groups[len(groups) // 3:]                     # skip the first pass (buffers grow, code objects load)
span = np.array([(g[-1][1] - g[0][0]) / 1e3 for g in groups])
busy = np.array([sum(e - s for s, e, _ in g) / 1e3 for g in groups])
print(f"# {len(groups)} registrations: first kernel start -> last kernel end {np.median(span):.1f} us (median), kernels busy {np.median(busy):.1f} us, "
      f"kernels per registration {np.median([len(g) for g in groups]):.0f}")
g = groups[len(groups) // 2]
t0 = g[0][0]
for s, e, n in g:
    print(f"  +{(s - t0) / 1e3:7.1f} us  {(e - s) / 1e3:6.1f} us  {n}")
