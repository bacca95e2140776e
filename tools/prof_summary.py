#!/usr/bin/env python3
"""Condense a rocprofv3 CSV output directory into a small text summary
(kernel-trace stats and/or per-kernel PMC averages).  Usage:
    python tools/prof_summary.py <rocprof_dir> [kernel-name-substring ...]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    pats = sys.argv[2:] or ["k_"]
    keep = lambda name: any(p in name for p in pats)
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        print(f"# {os.path.relpath(f, d)}")
        with open(f) as fh:
            rd = csv.DictReader(fh)
            print(",".join(rd.fieldnames))
            for r in rd:
                if keep(r.get("Name", "")):
                    print(",".join(r[k] for k in rd.fieldnames))
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0, None])
        with open(f) as fh:
            for r in csv.DictReader(fh):
                n = r["Kernel_Name"]
                if not keep(n):
                    continue
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                a = agg[n]
                a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
                a[4] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
                        r.get("Scratch_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
        print(f"# {os.path.relpath(f, d)}: name,calls,total_us,avg_us,min_us,max_us,(vgpr,agpr,sgpr,lds,scratch,wg,last_grid)")
        for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"{n},{a[0]},{a[1]:.1f},{a[1] / a[0]:.2f},{a[2]:.2f},{a[3]:.2f},{a[4]}")
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        with open(f) as fh:
            for r in csv.DictReader(fh):
                n = r["Kernel_Name"]
                if not keep(n):
                    continue
                c = agg[n][r["Counter_Name"]]
                c[0] += 1; c[1] += float(r["Counter_Value"])
        print(f"# {os.path.relpath(f, d)}: kernel,counter,dispatches,sum,avg_per_dispatch")
        for n, cs in agg.items():
            for cn, (k, v) in sorted(cs.items()):
                print(f"{n},{cn},{k},{v:.0f},{v / k:.1f}")


if __name__ == "__main__":
    main()
