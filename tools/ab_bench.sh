#!/bin/bash
# A/B of builds / settings on one box with one case cache, alternating, two rounds:
#   tools/ab_bench.sh <out_dir> "label|lib.so|extra bench.py args|ENV=...|case" ...
# (variants naming the same `case` share one generated workload: they must then differ in library / environment only)
# (empty lib: the in-tree lio-slam_amd/libliogpu.so).  Prints label, value (reg/s), roofline.ms_per_launch per run.
out=$1; shift
mkdir -p "$out"
for round in 1 2; do
  for v in "$@"; do
    IFS='|' read -r label lib extra envs case <<< "$v"
    env LIOGPU_LIB=$lib $envs python bench.py --no-cpu --no-extras --case-cache /tmp/ab_case_${case:-default}.npz $extra > "$out/bench_${label}_$round.json" 2> "$out/bench_${label}_$round.err" || exit 1
    python - "$out/bench_${label}_$round.json" "$label" <<'PY' | tee -a "$out/ab.txt"
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["value"]), round(d["roofline"]["ms_per_launch"], 5), round(d.get("map_build_ms", 0), 3), d.get("map_rows"))
PY
  done
done
