#!/usr/bin/env python3
"""One mapping callback as the patched node issues it: extractCloud from 200 resident keyframes (lio_assemble_map_resident),
downsampleCurrentScan + scan2MapOptimization from the RAW deskewed 64x1800 cloud as one device chain
(lio_s2m_register_raw), transformUpdate.  Wall time per phase; run under `rocprofv3 --kernel-trace` and summarise with
tools/prof_summary.py for the per-kernel split.   python tools/callback_trace.py"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("lio-slam_amd")
synth = importlib.import_module("lio-slam_amd.synth")
N = 16
LEAF_S = float(sys.argv[1]) if len(sys.argv) > 1 else 0.4       # mappingSurfLeafSize
LEAF_M = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5       # surroundingKeyframeMapLeafSize
case = synth.make_case("hdl64", n_keyframes=200, seed=synth.BASE_SEED, n_queries=N, device="cuda", workers=8, n_raw=N,
                       scan_leaf=LEAF_S, map_leaf=LEAF_M)
kc = [np.concatenate([c, np.zeros((len(c), 1), np.float32)], 1) for c, _ in case["keyframes"]]
kp = np.stack([p for _, p in case["keyframes"]])
PIPE = int(sys.argv[3]) if len(sys.argv) > 3 else 0             # cfg.pipeline of the node's handle (0 = auto)
node = pkg.ScanToMap(pipeline=PIPE)
store = pkg.KeyframeStore()
ids = [store.add(c) for c in kc]
recs = []
for q in case["queries"]:
    r = q["raw"]
    rec = np.zeros((len(r), 8), np.float32)
    rec[:, :3], rec[:, 3], rec[:, 4] = r[:, :3], 1.0, r[:, 3]
    recs.append(rec)
lay = pkg.PC2Layout(point_step=32, off_x=0, off_intensity=16, off_ring=-1, off_time=-1, pin_host=0)


def loop():
    ta = tr = 0.0
    nds = 0
    for i, q in enumerate(case["queries"]):
        t0 = time.perf_counter()
        store.assemble(ids, kp, LEAF_M, s2m=node, want_output=False)
        t1 = time.perf_counter()
        p, res, rc, nd = node.downsampleAndScan2MapOptimization(recs[i], len(recs[i]), lay, LEAF_S, q["pose_init"])
        pkg.transform_update(p)
        t2 = time.perf_counter()
        ta += t1 - t0; tr += t2 - t1; nds += nd
    return 1e3 * ta / N, 1e3 * tr / N, nds / N


loop()
a, r, nd = loop()
print(f"callback (leaf {LEAF_S} / {LEAF_M}): assemble {a:.3f} ms + downsample+register+transformUpdate {r:.3f} ms = {a + r:.3f} ms; raw {np.mean([len(x) for x in recs]):.0f} points -> N_s {nd:.0f}; "
      f"pipeline {node.profile().pipeline}, fallbacks {node.profile().persist_fallbacks}")
