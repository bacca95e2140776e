#!/usr/bin/env python3
"""Per-GN-iteration kernel durations of the last timed step from a rocprofv3 --kernel-trace CSV directory:
    python tools/split_trace.py <rocprof_dir> [launches_per_step]
Prints, for the last `launches_per_step` GN iterations, the duration of every k_s2m_* dispatch in order."""
import csv, glob, os, sys
d = sys.argv[1]
per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "k_s2m_" in r["Kernel_Name"] and "init_state" not in r["Kernel_Name"] and "pack_summary" not in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0]))
rows.sort()
names = sorted({r[2] for r in rows})
kinds = len(names)
tail = rows[-per_step * kinds:]
print("# start_us(rel) " + " | ".join(names) + " | gap_to_next_us")
t0 = tail[0][0]
for i in range(0, len(tail), kinds):
    grp = tail[i:i + kinds]
    durs = {n: 0.0 for n in names}
    for s, e, n in grp:
        durs[n] += (e - s) / 1e3
    nxt = tail[i + kinds][0] if i + kinds < len(tail) else grp[-1][1]
    span = (grp[-1][1] - grp[0][0]) / 1e3
    print(f"{(grp[0][0] - t0) / 1e3:9.1f}  " + " | ".join(f"{durs[n]:8.1f}" for n in names) + f" | span {span:8.1f} busy {sum(durs.values()):8.1f}")
