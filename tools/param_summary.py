#!/usr/bin/env python3
"""One line per parameter set of tools/param_sets.sh: sizes, value, ms per launch, roofline fraction, L2 hit rate, HBM
traffic against the algorithmic bytes.   python tools/param_summary.py <dir> <tag>"""
import json
import os
import sys


def avg(path, counter):
    if not os.path.exists(path):
        return None
    for line in open(path):
        if "k_s2m_iterate<1, false, false>" in line and "," + counter + "," in line:
            return float(line.rstrip().rsplit(",", 1)[1])
    return None


def main():
    d, tag = sys.argv[1], sys.argv[2]
    try:
        j = json.loads(open(os.path.join(d, f"bench_{tag}.json")).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(f"{tag}: no bench line ({e})")
        return
    c, r = j["config"], j["roofline"]
    rows = j.get("map_rows", {})
    kdiv, tables = c.get("kernel", {}).get("cell_div", 2), rows.get("tight_tables", 0)
    hit, miss = avg(os.path.join(d, f"pmc_TCC_{tag}.txt"), "TCC_HIT_sum"), avg(os.path.join(d, f"pmc_TCC_{tag}.txt"), "TCC_MISS_sum")
    fs, ws = avg(os.path.join(d, f"pmc_FETCH_{tag}.txt"), "FETCH_SIZE"), avg(os.path.join(d, f"pmc_FETCH_{tag}.txt"), "WRITE_SIZE")
    kt = None
    p = os.path.join(d, f"kernel_stats_{tag}.txt")
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("void k_s2m_iterate<1, false, false>(LioIterParams),") and "('" in line:
                f = line.split(",")
                kt = (int(f[3]), float(f[5]))        # calls, avg_us
    traffic = (2 * fs + ws) * 1024 if fs is not None and ws is not None else None      # KB -> bytes, gfx950 FETCH_SIZE correction x2
    out = {
        "set": tag, "leaf_scan": c["mappingSurfLeafSize"], "leaf_map": c["surroundingKeyframeMapLeafSize"], "scans_per_step": c["scans_per_step"],
        "N_s_mean": round(c["N_s_mean"]), "N_m": c["N_m"], "rows_MB": round(c["N_m"] * ((2 * kdiv + 1) ** 2 + 9 * tables) * 16 / 1e6, 1), "x_sub": rows.get("x_sub"), "tight_tables": tables, "first_try": rows.get("first_try"),
        "points_per_occupied_cell": rows.get("points_per_occupied_cell"), "gn_iters_mean": round(c["gn_iters_mean"], 2),
        "value_reg_per_s": round(j["value"]), "ms_per_step": round(j["ms_per_step"], 3),
        "ms_per_launch_hip_events": round(r["ms_per_launch"], 4), "ms_per_launch_rocprofv3": round(kt[1] / 1e3, 4) if kt else None,
        "algorithmic_MB_per_launch": round(r["algorithmic_bytes_per_launch"] / 1e6, 2), "frac_hbm": round(r["frac"], 4),
        "l2_hit": round(hit / (hit + miss), 4) if hit is not None and (hit + miss) > 0 else None,
        "hbm_traffic_MB_per_launch": round(traffic / 1e6, 2) if traffic is not None else None,
        "traffic_over_algorithmic": round(traffic / r["algorithmic_bytes_per_launch"], 3) if traffic and r["algorithmic_bytes_per_launch"] else None,
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
