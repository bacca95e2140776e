#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of k_s2m_iterate (cfg.profile=2 stamps).
   python tools/phase_clock.py /tmp/case.npz [ppt] [lds] [sort] [max_iters] [cell_div] [key=value ...]   (extra S2MConfig fields)"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("lio-slam_amd")
z = np.load(sys.argv[1])
ppt, lds, srt = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 1), (3, 0), (4, 0)))
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 2
cdiv = int(sys.argv[6]) if len(sys.argv) > 6 else 2
extra = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[7:] if "=" in a}
lens = z["lens"][:512]                               # one batch of the bench's case cache
offs = np.concatenate([[0], np.cumsum(lens)])
cat = z["scans"]                                     # (an NpzFile re-reads the member on every access: take it ONCE)
scans = [np.ascontiguousarray(cat[offs[i]:offs[i + 1]]) for i in range(len(lens))]
del cat
s2m = pkg.ScanToMap(profile=2, kernel_variant=ppt, use_lds=lds, sort_scan=srt, max_iters=iters, lookahead=0, cell_div=cdiv, **extra)
s2m.set_map(z["map"]); s2m.batch_upload(scans); s2m.batch_set_poses(z["poses0"][:len(lens)])
s2m.batch_run(); s2m.batch_results(False)
st = s2m.debug_stamps().astype(np.float64)
ok = st[:, :, 5] > 0
names = ["load+transform(+stage)", "kNN (pp0)", "fit+jacobian (+other pp)", "wave reduce", "barrier", "store+arrive", "tail"]
d = np.diff(st, axis=2)
print(f"ppt={ppt} lds={lds} sort={srt} cell_div={cdiv} iters={iters} {extra}: blocks={len(st)} waves timed={int(ok.sum())}")
for k, nme in enumerate(names[:5]):
    v = d[:, :, k][ok]
    print(f"  {nme:28s} mean {v.mean():9.0f}  p50 {np.median(v):9.0f}  p95 {np.percentile(v, 95):9.0f} cycles")
w0 = st[:, 0, 6] > 0
v = (st[:, 0, 6] - st[:, 0, 5])[w0]
print(f"  {'store+arrive (wave 0)':28s} mean {v.mean():9.0f}  p50 {np.median(v):9.0f}  p95 {np.percentile(v, 95):9.0f} cycles")
tot = (st[:, :, 5] - st[:, :, 0])[ok]
print(f"  total entry->barrier exit    mean {tot.mean():9.0f}  p50 {np.median(tot):9.0f}")
print(f"  launch span: {(st[:, 0, 6][w0].max() - st[:, :, 0][ok].min()):.0f} ticks")
