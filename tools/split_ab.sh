#!/bin/bash
# A/B of the split pipeline's knobs under rocprofv3 --kernel-trace: per-iteration kernel durations of the last step.
#   bash tools/split_ab.sh "<env assignments>" ...     e.g.  bash tools/split_ab.sh "LIO_SPLIT_SORT=0" "LIO_SPLIT_SORT=1"
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
CASE=/tmp/c512.npz
[ -f $CASE ] || python bench.py --steps 1 --warmup 0 --no-cpu --case-cache $CASE > /dev/null 2>&1
i=0
for envs in "$@"; do
    i=$((i+1))
    rm -rf /tmp/prof/ab$i
    env $envs rocprofv3 --kernel-trace --output-format csv -d /tmp/prof/ab$i -o run -- python bench.py --steps 3 --warmup 2 --no-cpu --pipeline ${PIPE:-2} --case-cache $CASE > /tmp/ab$i.json 2> /tmp/ab$i.err
    echo "== $envs"
    python tools/split_trace.py /tmp/prof/ab$i
done
