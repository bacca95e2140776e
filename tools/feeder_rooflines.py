#!/usr/bin/env python3
"""Roofline table of the feeder kernels from the rocprofv3 kernel stats of tools/profile_round.sh (round-2 verdict, next #7):
per kernel group the GPU time per call, the algorithmic bytes of SURVEY 8(d), the achieved rate and its fraction of the
8 TB/s HBM peak.  These kernels touch 2-21 MB once, i.e. 0.3-3 us at the peak: launch latency is their floor, and the table
says so with numbers.

   python tools/feeder_rooflines.py profiles/r03_feeders_kernel_stats.txt profiles/r03_map_assembly_kernel_stats.txt
"""
import re
import sys

PEAK = 8000.0  # GB/s


def parse(path):
    rows, sec = {}, False
    for line in open(path):
        if line.startswith("# runc_kernel_trace.csv"):
            sec = True
            continue
        if not sec or line.startswith("#"):
            continue
        m = re.match(r"^(.*),(\d+),([\d.]+),([\d.]+),([\d.]+),([\d.]+),\(", line)
        if not m:
            continue
        name = m.group(1)
        short = "k_s2m_persist" if "k_s2m_persist" in name else name.replace("void ", "").split("(")[0].split("<")[0]
        if short.startswith("k_"):
            c, t = rows.get(short, (0, 0.0))
            rows[short] = (c + int(m.group(2)), t + float(m.group(3)))
    return rows


def group(rows, prefixes, calls):
    t = sum(v[1] for k, v in rows.items() if k.startswith(tuple(prefixes)))
    n = sum(v[0] for k, v in rows.items() if k.startswith(tuple(prefixes)))
    return t / calls, n / calls


def line(name, us, launches, mb, note=""):
    gbs = mb * 1e6 / (us * 1e-6) / 1e9 if us > 0 else 0.0
    print(f"| {name} | {us:.1f} | {launches:.0f} | {mb:.2f} | {gbs:.0f} | {100 * gbs / PEAK:.1f} % | {note} |")


def main():
    feed = parse(sys.argv[1])
    asm = parse(sys.argv[2]) if len(sys.argv) > 2 else {}
    n_raw, n_kf, n_map = 114612, 1290461, 67323          # tools/prof_feeders.py inputs (printed in *_feeders_run.txt)
    reps = 6                                              # 1 warm-up + 5 timed calls of every entry point in prof_feeders.py
    print("| kernels | GPU us per call | launches | algorithmic MB | GB/s | of 8 TB/s | what one call is |")
    print("|---|---|---|---|---|---|---|")
    us, nl = group(feed, ["k_deskew_"], 2 * reps)
    line("K1 `k_deskew_flags/_scan/_emit`", us, nl, n_raw * 44 / 1e6, "projectPointCloud + deskewPoint of a 64x1800 sweep (28 B in + 16 B out per point)")
    us, nl = group(feed, ["k_curvature"], reps)
    line("K2 `k_curvature`", us, nl, n_raw * 8 / 1e6, "calculateSmoothness, 8 B per point")
    us, nl = group(feed, ["k_feat_"], reps)
    line("FE `k_feat_*`", us, nl, n_raw * 28 / 1e6, "occlusion + feature selection + per-ring voxel filter (28 B per point)")
    us, nl = group(feed, ["k_ri_"], reps)
    line("A4 `k_ri_*` (extension)", us, nl, n_raw * 60 / 1e6, "range image + cloudExtraction (28 B in + 32 B out per point)")
    if asm:
        calls = 23                                        # tools/assemble_trace.py: 3 warm-up + 20 timed assemblies
        us, nl = group(asm, ["k_kf_transforms", "k_transform_clouds", "k_bbox_reduce"], calls)
        line("K6 `k_transform_clouds_bbox` (+ poses, box)", us, nl, n_kf * 32 / 1e6, "200 keyframe clouds into the world frame (16 B in + 16 B out per point)")
        us, nl = group(asm, ["k_vsort_", "k_vox_"], calls)
        line("K7 `k_vsort_*`", us, nl, (n_kf * 24 + n_map * 16) / 1e6, "voxel filter 1.29 M -> 67 k points (16 B + key + index per point, 16 B per voxel)")
        us, nl = group(asm, ["k_map_", "k_s2_", "k_xyzi4_to_soa"], calls)
        line("grid build `k_map_*`", us, nl, n_map * (32 + 400) / 1e6, "hash grid + 25x replicated rows of the new local map")


if __name__ == "__main__":
    main()
