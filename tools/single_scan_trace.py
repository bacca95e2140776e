#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV directory of tools/latency_sweep.py: duration of the single-scan k_s2m_iterate
launches and the idle gap between two consecutive ones of a registration (what a persistent one-launch loop could
remove, to be compared with the price of a grid-wide barrier).   python tools/single_scan_trace.py <dir>"""
import csv, glob, os, sys
import numpy as np
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
dur, gap = [], []
for (s0, e0, n0), (s1, e1, n1) in zip(rows[:-1], rows[1:]):
    if "k_s2m_iterate" in n0:
        dur.append((e0 - s0) / 1e3)
        if "k_s2m_iterate" in n1 and (s1 - e0) < 100000:      # next GN iteration of the same registration
            gap.append((s1 - e0) / 1e3)
dur, gap = np.array(dur), np.array(gap)
print(f"# single-scan k_s2m_iterate launches: {len(dur)}; duration mean {dur.mean():.1f} us, median {np.median(dur):.1f}, p10 {np.percentile(dur, 10):.1f}, p90 {np.percentile(dur, 90):.1f}")
print(f"# (the trace reports consecutive launches of one stream back to back, so the boundary between two GN iterations is not visible in it;")
print("#  the guide's price list is used for it)")
print("# (MI355X_MICROARCH.md price list: dependent kernel boundary 1.45-1.9 us; grid barrier 4.1-7.4 us at one workgroup per CU)")
