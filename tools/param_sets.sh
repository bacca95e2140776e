#!/bin/bash
# The reference's own parameter sets and a map that spills the caches (round-2 verdict, next #4), each as: the bench line
# (streamed value + roofline pass), then the roofline pass alone under rocprofv3 --kernel-trace --stats and two PMC passes
# (TCC hits / misses, FETCH_SIZE).  Run on the GPU box from the repo root:  bash tools/param_sets.sh [set ...]
#   default  0.4 / 0.5   lio_sam_default.yaml:56,71 (the headline)
#   jeep     0.2 / 0.5   config/jeep.yaml:99,114, config/m1.yaml:88
#   livox    0.15 / 0.3  config/lio_sam_livox.yaml:56,71
#   6t       0.01 / 0.5  config/6t.yaml:112,127: the leaf overflows PCL's voxel index, the scan is NOT downsampled (N_s ~ 115 k)
#   dense    0.2 / 0.1, obstacle density 0.06: N_m >= 0.5 M, the 25x replicated rows (>= 200 MB) no longer fit L2 + Infinity Cache
#   dense_k1 / dense_k3   the same with cfg.cell_div = 1 / 3 (9x / 49x replicated rows)
#   long     1000 keyframes along a 1 km street with 6x the obstacles, default leaves: a LARGE map at the usual point spacing
#   km3      3000 keyframes along a 3 km street, default leaves and obstacle density: N_m ~ 1 M at the USUAL point spacing, the
#            25x rows (~400 MB) no longer fit the 256 MB Infinity Cache; km3_k1: the same with cell_div 1 (9x rows, ~150 MB)
#   whatif   timing only (WRONG results by construction): the 5-NN gate shrunk to 0.8 m / 0.667 m, the upper bound of what a
#            second, tighter row table could win once a search bound exists (next #5)
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/param_r03
mkdir -p $OUT
SETS=${@:-default jeep livox 6t dense whatif}
one() { # tag, bench args...
    local tag=$1; shift
    local CASE=/tmp/case_$tag.npz
    local B="python bench.py --no-cpu --no-extras --case-cache $CASE --label $tag $@"
    timeout -k 10 900 $B --steps 8 --warmup 3 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
    echo "$tag bench rc=$?"
    local R="$B --steps 6 --warmup 2 --roofline-pass-only"
    rm -rf /tmp/prof/p_$tag
    echo "$tag bench done" >> $OUT/progress.log
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/p_$tag/kt -o runc -- $R > /dev/null 2> /tmp/p_$tag.err
    python tools/prof_summary.py /tmp/prof/p_$tag/kt k_s2m k_scan k_map > $OUT/kernel_stats_$tag.txt
    echo "$tag kernel trace done" >> $OUT/progress.log
    timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d /tmp/prof/p_$tag/tcc -- $R > /dev/null 2>&1
    python tools/prof_summary.py /tmp/prof/p_$tag/tcc k_s2m_iterate > $OUT/pmc_TCC_$tag.txt
    echo "$tag TCC pass done" >> $OUT/progress.log
    # (FETCH_SIZE and WRITE_SIZE in ONE pass never finished on this pool: one derived counter per pass, as in tools/profile_round.sh)
    timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof/p_$tag/fs -- $R > /dev/null 2>&1
    echo "$tag FETCH pass rc=$?" >> $OUT/progress.log
    timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof/p_$tag/ws -- $R > /dev/null 2>&1
    echo "$tag WRITE pass rc=$?" >> $OUT/progress.log
    python tools/prof_summary.py /tmp/prof/p_$tag/fs k_s2m_iterate > $OUT/pmc_FETCH_$tag.txt
    python tools/prof_summary.py /tmp/prof/p_$tag/ws k_s2m_iterate >> $OUT/pmc_FETCH_$tag.txt
    rm -f $CASE
    python tools/param_summary.py $OUT $tag >> $OUT/summary.txt
    tail -1 $OUT/summary.txt
}
for s in $SETS; do
    case $s in
    default) one default --leaf-scan 0.4 --leaf-map 0.5 ;;
    jeep)    one jeep --leaf-scan 0.2 --leaf-map 0.5 --batch 256 --batches 2 ;;
    livox)   one livox --leaf-scan 0.15 --leaf-map 0.3 --batch 128 --batches 2 ;;
    6t)      one 6t --leaf-scan 0 --leaf-map 0.5 --batch 32 --batches 2 ;;
    dense)   one dense --leaf-scan 0.2 --leaf-map 0.1 --density 0.06 --batch 128 --batches 2 ;;
    dense_k1) one dense_k1 --leaf-scan 0.2 --leaf-map 0.1 --density 0.06 --batch 128 --batches 2 --celldiv 1 ;;
    dense_k3) one dense_k3 --leaf-scan 0.2 --leaf-map 0.1 --density 0.06 --batch 128 --batches 2 --celldiv 3 ;;
    livox_k3) one livox_k3 --leaf-scan 0.15 --leaf-map 0.3 --batch 128 --batches 2 --celldiv 3 ;;
    long)    one long --keyframes 1000 --density 0.12 --batch 256 --batches 2 ;;
    km3)     one km3 --keyframes 3000 --batch 256 --batches 2 ;;
    km3_k1)  one km3_k1 --keyframes 3000 --batch 256 --batches 2 --celldiv 1 ;;
    long_k1) one long_k1 --keyframes 1000 --density 0.12 --batch 256 --batches 2 --celldiv 1 ;;
    whatif)  one whatif_gate0.8 --maxsq 0.64 ; one whatif_gate0.667 --maxsq 0.4444 ;;
    esac
done
