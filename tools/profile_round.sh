#!/bin/bash
# Profiling session behind profiles/r03_*: the bench's roofline pass under rocprofv3, one pass per counter group (kernel
# trace and PMC are never combined), then the feeders (K1/K2/K7/FE/range image), the local-map assembly and the per-callback
# chain.  Run on the GPU box from the repo root:   bash tools/profile_round.sh [main|feeders|all]
# Summaries land in gpurun_out/prof_r03/; copy the ones to be kept into profiles/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
WHAT=${1:-all}
OUT=gpurun_out/prof_r03
mkdir -p $OUT
CASE=/tmp/case8.npz
run_pmc() { # tag, out-prefix, kernel filters (quoted, space separated), counters..., then "--" and the program
    local tag=$1 prefix=$2 filt=$3; shift 3
    local ctr=()
    while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
    shift
    rm -rf /tmp/prof/$tag
    timeout -k 10 400 rocprofv3 --pmc "${ctr[@]}" --output-format csv -d /tmp/prof/$tag -- "$@" > /tmp/$tag.out 2>&1
    echo "$tag rc=$?"
    python tools/prof_summary.py /tmp/prof/$tag $filt > $OUT/${prefix}pmc_$tag.txt
}
run_trace() { # tag, out file, kernel filters, program...
    local tag=$1 outf=$2 filt=$3; shift 3
    rm -rf /tmp/prof/$tag
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/$tag -o runc -- "$@" > $OUT/$tag.stdout 2> /tmp/$tag.err
    echo "$tag rc=$?"
    python tools/prof_summary.py /tmp/prof/$tag $filt > $OUT/$outf
}
if [ "$WHAT" = main ] || [ "$WHAT" = all ]; then
    python bench.py --steps 1 --warmup 1 --no-cpu --no-extras --case-cache $CASE > /dev/null 2> $OUT/gen.log   # writes the case cache
    # the roofline pass of the default bench as the ONLY timed region: the GN loop of one pre-sorted resident 512-scan batch with
    # nothing else in flight (in the streamed region two batches' launch loops overlap on purpose: durations there measure sharing)
    B="python bench.py --steps 16 --warmup 4 --no-cpu --no-extras --roofline-pass-only --case-cache $CASE"
    run_trace kt rocprofv3_kernel_stats.txt "k_" $B
    cp $OUT/kt.stdout $OUT/bench_under_rocprof.json
    run_pmc FETCH_SIZE "" "k_s2m k_scan" FETCH_SIZE -- $B
    run_pmc WRITE_SIZE "" "k_s2m k_scan" WRITE_SIZE -- $B
    run_pmc SQ "" "k_s2m k_scan" SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -- $B
    run_pmc TCC "" "k_s2m k_scan" TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -- $B
    run_pmc SQ2 "" "k_s2m k_scan" SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE GRBM_TA_BUSY -- $B
    python tools/issue_roofline.py $OUT > $OUT/issue_roofline.json
fi
if [ "$WHAT" = feeders ] || [ "$WHAT" = all ]; then
    FEED=/tmp/feed.npz
    python tools/prof_feeders.py gen $FEED > $OUT/feeders_gen.log 2>&1
    F="python tools/prof_feeders.py run $FEED"
    run_trace feeders feeders_kernel_stats.txt "k_" $F
    cp $OUT/feeders.stdout $OUT/feeders_run.txt
    run_pmc feeders_FETCH feeders_ "k_" FETCH_SIZE -- $F
    run_pmc feeders_WRITE feeders_ "k_" WRITE_SIZE -- $F
    A="python tools/assemble_trace.py"
    run_trace assemble map_assembly_kernel_stats.txt "k_" $A
    run_pmc assemble_FETCH assemble_ "k_" FETCH_SIZE -- $A
    run_pmc assemble_WRITE assemble_ "k_" WRITE_SIZE -- $A
    C="python tools/callback_trace.py"
    run_trace callback callback_kernel_stats.txt "k_" $C
    cp $OUT/callback.stdout $OUT/callback_run.txt
fi
grep -h "k_s2m_iterate" $OUT/rocprofv3_kernel_stats.txt $OUT/pmc_*.txt 2>/dev/null | cut -c1-200
