#!/usr/bin/env python3
"""The roofline of the resource that actually binds k_s2m_iterate (round-2 verdict, next #5): vector-ALU ISSUE, with the
texture addresser next to it -- from the PMC passes of tools/profile_round.sh (one counter group per run).

  issue cycles of the VALU per SIMD   = SQ_ACTIVE_INST_VALU x 4 / 1024      (the SQ counts quad-cycles, summed over all SIMDs)
  cycles of the launch                = GRBM_GUI_ACTIVE / 8                  (rocprofv3 reports the sum over the 8 XCDs)
  frac                                = the first / the second
  ta_busy                             = GRBM_TA_BUSY / GRBM_GUI_ACTIVE

Prints one JSON object (kept as profiles/r03_issue_roofline.json, read by bench.py as `roofline_issue`).
Usage: python tools/issue_roofline.py gpurun_out/prof_r03"""
import json
import os
import sys


def counters(path, kernel):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        if kernel in line and "," in line:
            f = line.rstrip().rsplit(",", 4)
            if len(f) == 5:
                try:
                    out[f[1]] = (int(f[2]), float(f[4]))
                except ValueError:
                    pass
    return out


def main():
    d = sys.argv[1]
    k = "k_s2m_iterate<1, false, false>"
    sq, sq2 = counters(os.path.join(d, "pmc_SQ.txt"), k), counters(os.path.join(d, "pmc_SQ2.txt"), k)
    if "SQ_ACTIVE_INST_VALU" not in sq or "GRBM_GUI_ACTIVE" not in sq2:
        print(json.dumps({"error": "counter files missing"}))
        return
    n_simd = 256 * 4
    valu_cycles_per_simd = sq["SQ_ACTIVE_INST_VALU"][1] * 4.0 / n_simd
    launch_cycles = sq2["GRBM_GUI_ACTIVE"][1] / 8.0
    out = {
        "bound": "valu_issue", "kernel": "k_s2m_iterate", "unit": "cycles per SIMD and launch",
        "achieved": valu_cycles_per_simd, "peak": launch_cycles, "frac": valu_cycles_per_simd / launch_cycles,
        "valu_instructions_per_launch": sq["SQ_INSTS_VALU"][1], "quad_cycles_per_valu_instruction": sq["SQ_ACTIVE_INST_VALU"][1] / sq["SQ_INSTS_VALU"][1],
        "texture_addresser_busy_frac": sq2["GRBM_TA_BUSY"][1] / sq2["GRBM_GUI_ACTIVE"][1],
        "vmem_read_instructions_per_launch": sq2.get("SQ_INSTS_VMEM_RD", (0, 0.0))[1],
        "wave_cycles_waiting_frac": (sq["SQ_WAIT_ANY"][1] / sq["SQ_WAVE_CYCLES"][1]) if "SQ_WAIT_ANY" in sq else None,
        "dispatches": sq["SQ_ACTIVE_INST_VALU"][0],
        "derivation": "SQ_ACTIVE_INST_VALU (quad-cycles, all SIMDs) x 4 / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs; separate rocprofv3 --pmc passes "
                      "of `bench.py --roofline-pass-only` (tools/profile_round.sh)",
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
